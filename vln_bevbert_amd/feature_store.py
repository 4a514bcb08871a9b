"""Device-resident store of the per-viewpoint grid features.

The reference reads, for every sample of every batch, the 12 x 14 x 14 CLIP patch features of the current viewpoint from
HDF5 (fp16 on disk), widens them to fp32, one-hot encodes the semantic ids in float64 and ships ~8 MB per sample to the
GPU (pretrain_src/data/dataset.py:110-118,397-440; 462 MB per step at batch 64).  R2R has 10 567 viewpoints:
12*196*768 fp16 = 3.6 MB each, 38 GB in all -- it fits in a corner of the MI355X's 288 GB.  So the store keeps

    rgbs    (N, 2352, 768) fp16      as on disk
    depths  (N, 12, 14, 14) fp32     stored / depth_scale, as on disk
    sems    (N, 2352)       uint8    class ids, as on disk

resident in HBM, a batch is described by B row numbers, and the splat kernel reads its points straight from the store
rows (``sample_rows`` of bevbert_bev_splat_mean): no per-step H2D traffic, no batch copy, and the fp16 -> fp32 widening
happens in the kernel's registers.
"""
import numpy as np
import torch


class GridFeatureStore:
    def __init__(self, keys, rgbs, depths, sems, device):
        """keys: N strings '<scan>_<viewpoint>' (the reference's HDF5 keys); rgbs (N, V, hw*hw | hw, hw, C) fp16/fp32;
        depths (N, V, hw, hw); sems (N, V, hw, hw) or (N, P) integer class ids."""
        keys = list(keys)
        N = len(keys)
        rgbs, depths, sems = (torch.as_tensor(np.asarray(t)) if not torch.is_tensor(t) else t for t in (rgbs, depths, sems))
        assert rgbs.shape[0] == N and depths.shape[0] == N and sems.shape[0] == N
        self.V, self.hw = int(depths.shape[1]), int(depths.shape[-1])
        self.C = int(rgbs.shape[-1])
        self.P = self.V * self.hw * self.hw
        self.row = {k: i for i, k in enumerate(keys)}
        assert len(self.row) == N, "duplicate keys"
        self.device = torch.device(device)
        self.rgbs = rgbs.reshape(N, self.P, self.C).to(self.device, torch.float16).contiguous()
        self.depths = depths.reshape(N, self.V, self.hw, self.hw).to(self.device, torch.float32).contiguous()
        self.sems = sems.reshape(N, self.P).to(self.device, torch.uint8).contiguous()

    def __len__(self):
        return len(self.row)

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in (self.rgbs, self.depths, self.sems))

    def rows(self, keys):
        """(B,) int32 device tensor of store rows for a list of '<scan>_<viewpoint>' keys (one small H2D copy)."""
        try:
            idx = [self.row[k] for k in keys]
        except KeyError as e:
            raise KeyError(f"viewpoint {e.args[0]!r} is not in the grid-feature store") from None
        return torch.tensor(idx, dtype=torch.int32).to(self.device, non_blocking=True)

    def attach(self, batch, keys):
        """Replace the per-batch grid tensors of a collated batch by references into the store:
        GlocalTextPathCMTPreTraining.lift_splat then reads 'grid_store' / 'grid_rows' instead of rgbs / depths / sems."""
        for k in ("rgbs", "depths", "sems"):
            batch.pop(k, None)
        batch["grid_store"] = self
        batch["grid_rows"] = self.rows(keys)
        return batch

    def gather(self, rows):
        """Materialised (rgbs fp16, depths, sems) of a batch -- what the zero-copy path avoids; used by the tests."""
        r = rows.long()
        return self.rgbs.index_select(0, r), self.depths.index_select(0, r), self.sems.index_select(0, r)
