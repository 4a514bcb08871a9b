"""Static-shape batches and hipGraph capture of the whole training step.

The reference's step is ~1 000 kernel launches issued one by one from Python (pretrain_src/train_r2r.py:247-313); at
batch 64 on an MI355X the GPU needs ~17 ms for them and the Python autograd + ctypes enqueue ~20 ms, so the step is
host-bound.  A HIP graph replays the whole step -- lift + splat, forward, backward (weight-gradient stream forked and
joined inside), gradient clipping, AdamW -- with one launch.  Three things make the step capturable:

  * every quantity that changes from step to step lives in DEVICE memory, not in launch arguments: the dropout salt
    (ops._Runtime / bevbert_set_step_salt) and the learning rate (ParamArena.set_lr);
  * everything the reference builds on the host INSIDE the forward -- positions of the masked tokens, the SAP logit
    fusion table, the global-map aggregation CSR -- is built by the loader (``StaticBatch``) next to its host->device
    copies; data-dependent row counts (masked tokens, supervised BEV cells) are padded to a fixed size with
    zero-weight rows, so the step has no host<->device synchronisation and fixed kernel shapes;
  * a ``StaticBatch`` owns preallocated device buffers of one shape bucket: the loader refills them in place
    (``load``), the graph captured on them is replayed.  Device-resident grid features
    (feature_store.GridFeatureStore) enter as row numbers, so a refill moves kilobytes, not the 462 MB of CLIP grid
    features the reference ships per step at batch 64.

``PretrainTrainer.step(task, static_batch)`` runs such a batch eagerly twice (library plans settle, buffers warm up),
captures the third step, and replays from then on.  Eager and replayed steps execute the same kernels with the same
arguments: losses and parameters agree bit for bit (tests/test_gpu_zz_streams.py).
"""
import numpy as np
import torch

from . import ops, synthetic
from .pretrain_cmt import sap_fusion_indices
from .vilmodel import gmap_csr_arrays

MLM_ROW_PAD = 256       # masked-token rows are padded to a multiple of this (15 % of B x L tokens +- a few dozen: a coarse
                        # step keeps consecutive batches in ONE shape bucket, i.e. on one captured graph)
SEM_ROW_PAD = 512       # supervised BEV cells (MaskSEM) likewise
_STAGE_PINNED = __import__("os").environ.get("BEVBERT_STAGE_PINNED", "1") == "1"      # A/B knob (StaticBatch._staged)
GMAP_PAD = 4            # global-map width G (batch max of the node counts) is rounded up to a multiple of this
# Round 5: ragged batches.  The reference pads every batch to ITS OWN maxima (pretrain_src/data/tasks.py:116-163), so the
# flat panorama count (sum of the path lengths, 64 draws of 1..7), the text width and the view count change from step to
# step -- one shape bucket (= one captured graph) per batch, i.e. no replays at all (r05 measurement: 64 buckets, 27 of
# them created inside a 33-step window, every step eager, 0.66 x the resident rate).  Each varying axis is rounded up:
# text to TXT_PAD tokens (beyond txt_lens: masked like the reference's own padding), panoramas to PANO_PAD dummy
# panoramas of ONE zero view (no segment of the global-map aggregation refers to them and their outputs feed nothing, so
# they contribute exact zeros to every gradient; one valid view keeps their softmax finite), views to VIEW_PAD.
# Dropout masks are a hash of (seed, launch counter, row-major element index): a padded axis moves the indices, so a padded
# batch draws DIFFERENT (equally distributed) masks than the same batch unpadded or than a run of rounds 1-4 with the same
# seed -- run-to-run reproducibility holds for a fixed set of pads, not across pad settings (all three = 1 restores the
# unpadded indices).  With dropout off the two are the same step (test_static_batch_step_equals_the_reference_api_step).
_env_int = lambda k, d: int(__import__("os").environ.get(k, d))
TXT_PAD = _env_int("BEVBERT_TXT_PAD", 16)
PANO_PAD = _env_int("BEVBERT_PANO_PAD", 32)
VIEW_PAD = _env_int("BEVBERT_VIEW_PAD", 4)


def _round_up(n, m):
    return (n + m - 1) // m * m


class StaticBatch:
    """Device buffers + loader-built index tensors of one batch of one task, in a fixed shape bucket."""

    def __init__(self, cfg, task, batch, device, grid_store=None, grid_keys=None):
        """``batch``: the collate output on the host (synthetic.collate / the reference's *_collate schema).
        ``grid_store`` + ``grid_keys``: draw the grid features from a device-resident GridFeatureStore instead of
        shipping ``rgbs`` / ``depths`` / ``sems`` (the keys name the samples' viewpoints)."""
        self.cfg, self.task, self.device = cfg, task, torch.device(device)
        self.graph = None           # set by PretrainTrainer once the step has been captured on these buffers
        self.loss_out = None
        self.eager_runs = 0
        # Object tokens (REVERIE / SOON: traj_obj_img_fts) are gathered with indices the forward derives on the HOST from
        # traj_step_lens / traj_vp_obj_lens (vilmodel._obj_tokens) and uploads inside the forward: a captured graph would
        # either refuse the pageable copy or freeze the indices of the batch it was captured on.  Such batches run eagerly.
        self.capturable = batch.get("traj_obj_img_fts") is None
        host = self._host_side(batch)
        self.signature = host["signature"]
        t = {}
        src = dict(batch)
        src.update(host["padded"])
        for k, v in src.items():
            if grid_store is not None and k in ("rgbs", "depths", "sems"):
                continue
            t[k] = v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v
            if k in synthetic.HOST_COPIES and torch.is_tensor(v):
                t[k + "_cpu"] = v
        t["gmap_csr"] = (ops.SegmentCSR(*host["csr"], self.device, capacity=host["csr_capacity"]), host["G"])
        st = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in host["static"].items()}
        t["_static"] = st
        if grid_store is not None:
            grid_store.attach(t, grid_keys)
        self.tensors = t
        self._attach_masks()
        self._refresh_counts()

    @staticmethod
    def plan(cfg, task, batch):
        """The host-side part alone (shape signature, padded tensors, index tables) -- what a loader computes to find
        the bucket of a batch before it touches any device buffer (loader.BucketManager)."""
        self = object.__new__(StaticBatch)
        self.cfg, self.task = cfg, task
        return self._host_side(batch)

    # -- host side: everything that is a Python loop over ids or a data-dependent count --------------------------
    def _host_side(self, batch):
        cfg, task = self.cfg, self.task
        B = batch["txt_ids"].shape[0]
        G = _round_up(int(batch["gmap_lens"].max()), GMAP_PAD)
        padded = {}
        for k in ("gmap_step_ids", "gmap_visited_masks", "gmap_pos_fts"):       # (B, G0, ...) -> (B, G, ...)
            v = batch[k]
            if v.shape[1] < G:
                pad = v.new_zeros((B, G - v.shape[1]) + tuple(v.shape[2:]))
                v = torch.cat([v, pad], 1)
            padded[k] = v
        pd = batch["gmap_pair_dists"]
        if pd.shape[1] < G:
            full = pd.new_zeros(B, G, G)
            full[:, :pd.shape[1], :pd.shape[2]] = pd
            pd = full
        padded["gmap_pair_dists"] = pd
        has_obj = batch.get("traj_obj_img_fts") is not None
        # ---- text width
        L = batch["txt_ids"].shape[1]
        Lp = _round_up(L, TXT_PAD)
        if Lp > L:
            for k, fill in (("txt_ids", 0), ("txt_labels", -1)):
                if batch.get(k) is not None:
                    padded[k] = torch.cat([batch[k], batch[k].new_full((B, Lp - L), fill)], 1)
        txt_shape = (B, Lp)
        # ---- panorama count and view count (object-token batches run eagerly anyway: left as they are)
        view_lens = batch["traj_vp_view_lens"]
        T0, V0 = batch["traj_view_img_fts"].shape[:2]
        Tp, Vp = (T0, V0) if has_obj else (_round_up(T0, PANO_PAD), _round_up(V0, VIEW_PAD))
        if (Tp, Vp) != (T0, V0):
            for k in ("traj_view_img_fts", "traj_loc_fts", "traj_nav_types", "traj_view_dep_fts"):
                v = batch.get(k)
                if v is None:
                    continue
                full = v.new_zeros((Tp, Vp) + tuple(v.shape[2:]))
                full[:T0, :V0] = v
                padded[k] = full
            view_lens = torch.cat([view_lens, view_lens.new_ones(Tp - T0)])
            padded["traj_vp_view_lens"] = view_lens
        lens = view_lens
        if batch.get("traj_vp_obj_lens") is not None:
            lens = lens + batch["traj_vp_obj_lens"]
        n_views = batch["traj_loc_fts"].shape[1] if has_obj else Vp        # objects: the joint [views | objects] token axis
        rowptr, idx, w, n_src, G = gmap_csr_arrays(list(batch["traj_step_lens"]), lens.tolist(), batch["traj_vpids"],
                                                  batch["traj_cand_vpids"], batch["gmap_vpids"], n_views, G)
        n_src = Tp * n_views                       # the dummy panoramas are source rows no segment points at
        static = {}
        # sequence masks and their additive fp32 forms (vilmodel.gen_seq_masks / neg_key_mask / the panorama encoder's
        # key_padding_mask): pure functions of the length vectors the loader holds on the host -- built here, shipped with the
        # batch, hung on the length tensors (``_attach_masks``); the step then has no arange / compare / cast launches for them
        def seq_masks(lens_host, width, name, inf=False):
            m = torch.arange(width)[None, :] < torch.as_tensor(lens_host).reshape(-1, 1)
            static[name + "_masks"] = m
            static[name + "_km"] = (1.0 - m.to(torch.float32)) * -10000.0
            if inf:
                static[name + "_km_inf"] = torch.zeros(m.shape, dtype=torch.float32).masked_fill(~m, float("-inf"))
        static["bev_nav_long"] = batch["bev_nav_masks"].to(torch.int64)      # index into LocalBEVEncoder.nav_type_embedding
        seq_masks(batch["txt_lens"], Lp, "txt")
        seq_masks(batch["gmap_lens"], G, "gmap")
        if not has_obj:
            seq_masks(view_lens, Vp, "pano", inf=True)
        sig = [task, B, txt_shape, (Tp, Vp) + tuple(batch["traj_view_img_fts"].shape[2:]), G]
        if batch.get("traj_obj_img_fts") is not None:      # buffers of a bucket must agree on the object-token layout too
            # (incl. the width of the joint [views | objects] token axis: the maximum of views + objects over the panoramas)
            sig.append(("obj", tuple(batch["traj_obj_img_fts"].shape), tuple(int(x) for x in batch["traj_step_lens"]),
                        int(n_views)))
        if task.startswith("mlm"):
            labels = padded.get("txt_labels", batch["txt_labels"]).reshape(-1)
            pos = torch.nonzero(labels != -1).squeeze(1)
            n = int(pos.numel())
            npad = _round_up(max(n, 1), MLM_ROW_PAD)
            static["mlm_n"] = n
            static["mlm_pos"] = torch.cat([pos, pos.new_zeros(npad - n)])
            static["mlm_targets"] = torch.cat([labels[pos], labels.new_zeros(npad - n)])
            static["mlm_valid"] = (torch.arange(npad) < n).to(torch.float32)
            sig.append(npad)
        elif task.startswith("sap"):
            K = batch["bev_cand_idxs"].shape[1]
            cand_vpids = [[None] + c[-1] for c in batch["traj_cand_vpids"]]
            vis = padded["gmap_visited_masks"].tolist()
            src, vis_c = sap_fusion_indices(batch["gmap_vpids"], vis, cand_vpids, G, K)
            static["sap_src"] = torch.from_numpy(src)
            static["sap_vis_c"] = torch.from_numpy(vis_c)
            # flat row numbers (into the (B * cells, H) BEV states) of the candidate cells and of the centre cell
            cells = cfg.bev_dim * cfg.bev_dim
            base = torch.arange(B, dtype=torch.int64)[:, None] * cells
            static["sap_cand_flat"] = (base + batch["bev_cand_idxs"].to(torch.int64)).reshape(-1).contiguous()
            static["sap_center_flat"] = (base[:, 0] + (cells - 1) // 2).contiguous()
            sig.append(K)
        elif task.startswith("masksem"):
            static["sem_cap"] = _round_up(max(1, int(batch["bev_mrc_masks"].sum())), SEM_ROW_PAD)
            sig.append(static["sem_cap"])
        elif task.startswith("sem"):
            static["sem_cap"] = B * cfg.bev_dim * cfg.bev_dim
        return {"padded": padded, "static": static, "csr": (rowptr, idx, w, n_src), "csr_capacity": 2 * n_src, "G": G,
                "signature": tuple(sig)}

    # -- in-place refill -----------------------------------------------------------------------------------------
    def load(self, batch, grid_keys=None, host=None):
        """Write another batch of the same shape bucket into these buffers (what a prefetching loader does with its
        preallocated device buffers); a captured graph keeps replaying on them.  Raises if the bucket differs.
        ``host``: the result of ``plan`` for this batch if the caller has it already."""
        if host is None:
            host = self._host_side(batch)
        if host["signature"] != self.signature:
            raise ValueError(f"batch of shape bucket {host['signature']} does not fit the buffers of {self.signature}")
        t = self.tensors
        src = dict(batch)
        src.update(host["padded"])
        for k, v in src.items():
            if torch.is_tensor(v):
                if k in t and torch.is_tensor(t[k]):
                    t[k].copy_(self._staged(k, v), non_blocking=True)
                if k + "_cpu" in t:
                    t[k + "_cpu"] = v
            elif k in t:
                t[k] = v
        t["gmap_csr"][0].update(*host["csr"][:3])
        st = t["_static"]
        for k, v in host["static"].items():
            if torch.is_tensor(v):
                st[k].copy_(self._staged("_static." + k, v), non_blocking=True)
            else:
                st[k] = v                      # host integers (row counts): they are part of the shape bucket or a
        if "grid_store" in t and grid_keys is not None:      # divisor the step reads from device memory (see below)
            t["grid_rows"].copy_(t["grid_store"].rows(grid_keys), non_blocking=True)
        self._refresh_counts()
        return self

    def _attach_masks(self):
        """Hang the loader-built masks on the device tensors the model derives them from (refills copy in place, so the
        attributes stay valid for the life of the buffer set)."""
        t, st = self.tensors, self.tensors["_static"]
        for lens_key, name in (("txt_lens", "txt"), ("gmap_lens", "gmap"), ("traj_vp_view_lens", "pano")):
            m = st.get(name + "_masks")
            if m is None or not torch.is_tensor(t.get(lens_key)):
                continue
            m._km = st[name + "_km"]
            if name + "_km_inf" in st:
                m._km_inf = st[name + "_km_inf"]
            t[lens_key]._seq_masks = m
        if torch.is_tensor(t.get("bev_nav_masks")) and "bev_nav_long" in st:
            t["bev_nav_masks"]._long = st["bev_nav_long"]

    def _staged(self, key, v):
        """A pageable host tensor goes through a PINNED staging buffer this buffer set owns (allocated once per key): the
        host -> device copy is then truly asynchronous, and nothing pins / unpins memory per batch -- hipHostMalloc /
        hipHostFree are slow and synchronise with the device, which a DataLoader's pin_memory thread pays for every batch.
        The staging buffer is rewritten by the next refill of THIS set, which the loader orders behind the step that read
        the previous contents (loader.BucketManager), i.e. behind the copy out of it."""
        if self.device.type != "cuda" or not _STAGE_PINNED or v.numel() == 0 or v.is_pinned():
            return v
        st = self.__dict__.setdefault("_stage", {})
        buf = st.get(key)
        if buf is None or buf.shape != v.shape or buf.dtype != v.dtype:
            buf = st[key] = torch.empty(v.shape, dtype=v.dtype).pin_memory()
        buf.copy_(v)
        return buf

    def _refresh_counts(self):
        """Host-known divisors of the mean (the number of real masked-token rows) live in a device scalar, so that a
        captured step picks up the count of the batch that currently sits in the buffers."""
        st = self.tensors["_static"]
        if "mlm_n" in st:
            if "mlm_n_dev" not in st:
                st["mlm_n_dev"] = torch.zeros((), dtype=torch.float32, device=self.device)
            st["mlm_n_dev"].fill_(float(st["mlm_n"]))


class GraphedStep:
    """One captured training step bound to the buffers of a StaticBatch."""

    def __init__(self, graph, loss):
        self.graph, self.loss = graph, loss
