"""Host-side mirror of pretrain_src/model/pretrain_cmt.py: GlocalTextPathCMTPreTraining.

Same constructor (``config`` with ``pretrain_tasks`` / ``sem_pred_token``), same ``forward(batch, task,
compute_loss=True)`` contract (lift_splat pops the seven grid inputs from / adds the five BEV tensors to a COPY of
the batch, exactly like the reference's defaultdict copy), same un-reduced loss vectors, same ``state_dict`` keys.

lift_splat is two kernel launches for the whole batch (ops.bev_lift_bin + ops.bev_splat_mean) instead of a
B-iteration Python loop with three D2H syncs per sample (bev_utils.py:390-423).
"""

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .config import BevBertConfig
from .vilmodel import BertOnlyMLMHead, GlocalTextPathCMT, ensure_arena, finalize, from_pretrained, gen_seq_masks

BEV_DIM = 21      # pretrain_cmt.py:16-17 (the config's bev_dim / bev_res override these defaults)
BEV_RES = 0.5


def bevpos_polar(dim, device):
    """bev_utils.py:39-58: (dim*dim, 3) = (cos, sin, dist / (dim/2)) of the cell centres, y axis flipped."""
    c = torch.linspace(0.5, dim - 0.5, dim, dtype=torch.float32)
    ry, rx = torch.meshgrid(c, c, indexing="ij")
    ry = -(ry - dim / 2)
    rx = rx - dim / 2
    dis = (ry ** 2 + rx ** 2) ** 0.5
    cos, sin = rx / dis, ry / dis
    cos[dis == 0] = 0
    sin[dis == 0] = 0
    return torch.stack([cos, sin, dis / (dim / 2)], -1).reshape(dim * dim, 3).to(device)


class _Head(nn.Module):
    """Linear - ReLU - LayerNorm - Linear (pretrain_cmt.py:34-71: RegionClassification / ClsPrediction / MulClsPrediction)."""

    def __init__(self, hidden_size, out_dim, input_size=None):
        super().__init__()
        input_size = hidden_size if input_size is None else input_size
        self.net = nn.Sequential(nn.Linear(input_size, hidden_size), nn.ReLU(), nn.LayerNorm(hidden_size, eps=1e-12),
                                 nn.Linear(hidden_size, out_dim))

    def forward(self, x):
        l0, ln, l3 = self.net[0], self.net[2], self.net[3]
        if x.is_cuda and l0.weight.shape[0] % 4 == 0:
            # bias + ReLU in one launch behind the GEMM (forward and backward: the bias gradient rides on the ReLU backward)
            h = ops.bias_relu(ops.linear(x.contiguous(), l0.weight), l0.bias)
        else:
            h = torch.relu(ops.linear(x, l0.weight, l0.bias))
        h = ops.layernorm(h, ln.weight, ln.bias, 1e-12)
        return ops.linear(h, l3.weight, l3.bias)


class ClsPrediction(_Head):
    def __init__(self, hidden_size, input_size=None):
        super().__init__(hidden_size, 1, input_size)


class MulClsPrediction(_Head):
    def __init__(self, hidden_size, input_size=None):
        super().__init__(hidden_size, 40, input_size)


class RegionClassification(_Head):
    def __init__(self, hidden_size, label_dim):
        super().__init__(hidden_size, label_dim)


def sap_fusion_indices(gmap_vpids, visited_host, cand_vpids, G, K):
    """Host half of the SAP logit fusion (pretrain_cmt.py:339-356; fine-tune twin vilmodel.py:852-871).

    Returns src (B,G) int64 into [local logits | bw | 0] and the (B,K) mask of local candidates that are visited
    nodes.  cand_vpids[i] includes the leading None ([stop])."""
    B = len(gmap_vpids)
    src = np.full((B, G), K + 1, dtype=np.int64)        # K+1 -> the zero slot
    vis_c = np.zeros((B, K), dtype=bool)
    src[:, 0] = 0                                       # fused[:, 0] += local[:, 0]
    for i in range(B):
        visited = {vp for vp, m in zip(gmap_vpids[i], visited_host[i]) if m}
        tmp = {}
        for j, vp in enumerate(cand_vpids[i]):
            if j > 0:
                if vp in visited:
                    vis_c[i, j] = True
                else:
                    tmp[vp] = j                         # later duplicates overwrite, as the dict does
        for j, vp in enumerate(gmap_vpids[i]):
            if j > 0 and vp not in visited:
                src[i, j] = tmp.get(vp, K)              # K -> the accumulated backtrack logit
    return src, vis_c


def fuse_sap_logits(global_logits, local_logits, src, vis_c):
    bw = torch.where(vis_c, local_logits, torch.zeros_like(local_logits)).sum(1, keepdim=True)
    ext = torch.cat([local_logits, bw, torch.zeros_like(bw)], 1)
    return global_logits + ext.gather(1, src)


def _host_rows(batch, key):
    """Host copy of a small per-sample tensor: '<key>_cpu' if the loader kept one, else a single D2H copy."""
    t = batch.get(key + "_cpu")
    if t is None:
        t = batch[key]
    return t.tolist() if torch.is_tensor(t) else t


import os as _os

_MLM_ROW_PAD = int(_os.environ.get("BEVBERT_MLM_ROW_PAD", "0"))      # 0 = off (the reference's exact row count)


class GlocalTextPathCMTPreTraining(nn.Module):
    def __init__(self, config):
        super().__init__()
        config = BevBertConfig.adopt(config)     # a PretrainedConfig built from configs/*_model.json drops in (train_r2r.py:102-113)
        self.config = config
        self.bert = GlocalTextPathCMT(config)
        self.drop_env = nn.Dropout(config.feat_dropout)          # pretrain_cmt.py:79; read through ``feat_dropout``
        if "mlm" in config.pretrain_tasks:
            self.mlm_head = BertOnlyMLMHead(config)
        if "mrc" in config.pretrain_tasks:
            self.obj_classifier = RegionClassification(config.hidden_size, config.obj_prob_size)
        if "sap" in config.pretrain_tasks:
            self.global_sap_head = ClsPrediction(config.hidden_size)
            self.local_sap_head = ClsPrediction(config.hidden_size)
            self.sap_fuse_linear = ClsPrediction(config.hidden_size, input_size=config.hidden_size * 2) \
                if config.glocal_fuse else None
        if "og" in config.pretrain_tasks:
            self.og_head = ClsPrediction(config.hidden_size)
        if "sem" in config.pretrain_tasks or "masksem" in config.pretrain_tasks:
            self.local_sem_head = MulClsPrediction(config.hidden_size)
            self.sem_pred_token = config.sem_pred_token
        self.init_weights()
        self.tie_weights()
        self._proj = None

    # -- initialisation / checkpoint ABI ---------------------------------------------------------
    def init_weights(self):
        """BERT init (transformers' _init_weights, initializer_range 0.02): Linear/Embedding N(0, 0.02), LN (1, 0)."""
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                m.weight.data.normal_(mean=0.0, std=0.02)
                if isinstance(m, nn.Linear) and m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.LayerNorm):
                m.weight.data.fill_(1.0)
                m.bias.data.zero_()

    def tie_weights(self):
        if "mlm" in self.config.pretrain_tasks:     # pretrain_cmt.py:109-112
            self.mlm_head.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, config=None, state_dict=None, **kwargs):
        """The call train_r2r.py:153-155 makes; see vilmodel.from_pretrained."""
        return from_pretrained(cls, pretrained_model_name_or_path, config, state_dict)

    def finalize(self, device, compute_dtype=torch.float32, residual=None):
        """Place parameters in the flat arena on ``device`` (fp32 masters + optional bf16 compute copy)."""
        return finalize(self, device, compute_dtype, residual)

    feat_dropout = property(lambda self: self.drop_env.p, lambda self, v: setattr(self.drop_env, "p", v))

    def set_dropout(self, p):
        """Every dropout probability of the model, attention-probability dropout of the panorama encoder included
        (tests use it to switch dropout off).  The reference's own ``set_dropout(model, p)`` (utils/misc.py:19-25)
        works on this model unchanged as well: each site keeps its p in an ``nn.Dropout`` child (vilmodel._p_of), and
        like there it leaves the float ``dropout`` of the panorama encoder's attention alone."""
        for m in self.modules():
            if isinstance(m, nn.Dropout):
                m.p = p
            if hasattr(m, "attn_drop_p"):
                m.attn_drop_p = p

    # -- lift + splat -------------------------------------------------------------------------------
    def _projector(self, device):
        if self._proj is None or self._proj[0].device != device:
            cfg = self.config
            self._proj = (ops.pixel_scale(cfg.grid_hw, device), bevpos_polar(cfg.bev_dim, device))
        return self._proj

    def lift_splat(self, batch):
        """pretrain_cmt.py:114-167.  Grid inputs come either as the reference's per-batch tensors (rgbs, depths, sems)
        or as rows of a device-resident feature_store.GridFeatureStore ('grid_store' + 'grid_rows', zero-copy)."""
        cfg = self.config
        T_c2w, T_w2c, S_w2c = batch.pop("T_c2w"), batch.pop("T_w2c"), batch.pop("S_w2c")
        bev_gpos_fts = batch.pop("bev_gpos_fts")
        store, rows = batch.pop("grid_store", None), batch.pop("grid_rows", None)
        if store is not None:
            feat, sem_in = store.rgbs, (store.sems if cfg.sem_classes > 0 else None)
            depths = store.depths.index_select(0, rows.long())          # 9.4 KB per sample
        else:
            rgbs, depths, sems = batch.pop("rgbs"), batch.pop("depths"), batch.pop("sems", None)
            feat = rgbs.reshape(rgbs.shape[0], -1, rgbs.shape[-1])
            sem_in = None if sems is None else (sems if sems.dim() == 2 else sems.reshape(feat.shape[0], feat.shape[1], -1))
        B = depths.shape[0]
        dev = depths.device
        dim, K = cfg.bev_dim, cfg.bev_dim * cfg.bev_dim
        pix, polar = self._projector(dev)
        cell, order, cell_start = ops.bev_lift_bin(depths, T_c2w, T_w2c, S_w2c, pix, dim, cfg.bev_res)
        cd = ops._compute(self.bert.local_encoder.bev_fts_embeddings[0].weight).dtype
        bev_fts, bev_sems, bev_sem_masks = ops.bev_splat_mean(feat, order, cell_start, K, out_dtype=cd, sems=sem_in,
                                                              n_classes=cfg.sem_classes, rows=rows)
        bev_pos_fts = torch.cat([bev_gpos_fts.expand(-1, K, -1), polar[None].expand(B, -1, -1)], dim=-1)
        batch.update({
            "bev_fts": bev_fts,
            "bev_masks": self._all_ones(B, K, dev),                                # pretrain_cmt.py:152
            "bev_pos_fts": bev_pos_fts,
            "bev_sems": bev_sems,
            "bev_sem_masks": None if bev_sem_masks is None else bev_sem_masks.bool(),
            "_bev_masks_all_ones": True,
            "_bev_cell": cell,
        })
        return batch

    def _all_ones(self, B, K, dev):
        """The all-ones BEV mask of pretrain_cmt.py:152, built once per shape (it is read-only: the kernels skip it, see
        ``_bev_masks_all_ones``) instead of filled every step."""
        c = self.__dict__.setdefault("_ones_cache", {})
        t = c.get((B, K, dev))
        if t is None:
            t = c[(B, K, dev)] = torch.ones(B, K, dtype=torch.bool, device=dev)
        return t

    def drop_feats(self, batch):
        # fp32 loader features leave the dropout in the compute dtype (the cast the input projections need anyway)
        cd = ops._compute(self.bert.local_encoder.bev_fts_embeddings[0].weight).dtype
        for k in ("traj_view_img_fts", "traj_obj_img_fts", "bev_fts"):
            if batch.get(k) is not None:
                batch[k] = ops.dropout(batch[k], self.feat_dropout, self.training, out_dtype=cd)
        return batch

    # -- dispatcher ---------------------------------------------------------------------------------
    _CMT_KEYS = ["txt_ids", "txt_lens", "traj_view_img_fts", "traj_obj_img_fts", "traj_loc_fts", "traj_nav_types",
                 "traj_step_lens", "traj_vp_view_lens", "traj_vp_obj_lens", "traj_vpids", "traj_cand_vpids",
                 "gmap_lens", "gmap_step_ids", "gmap_pos_fts", "gmap_pair_dists", "gmap_vpids",
                 "bev_fts", "bev_pos_fts", "bev_masks", "bev_nav_masks"]

    def _cmt_args(self, b):
        args = [b.get(k) for k in self._CMT_KEYS]
        if b.get("_bev_masks_all_ones"):
            args[18] = True          # lets the kernels skip an all-ones mask (see vilmodel._all_ones_to_none)
        return args

    def loss_mean(self, batch, task):
        """Scalar mean loss of the step (``loss.mean()`` of train_r2r.py:263).  For a static batch
        (static_step.StaticBatch: data-dependent row counts padded to a fixed size so that the step can live in a
        hipGraph) the padded rows carry zero weight and the mean is taken over the real rows only."""
        return self.forward(batch, task, compute_loss="mean")

    def forward(self, batch, task, compute_loss=True):
        """``compute_loss``: True / False as in the reference; "mean" (internal, see ``loss_mean``) -> scalar."""
        if not any(task.startswith(t) for t in ("mlm", "mrc", "sap", "og", "sem", "masksem")):
            raise ValueError("invalid task")
        ensure_arena(self)
        batch = dict(batch)     # the reference works on a defaultdict COPY (pretrain_cmt.py:170): the caller's dict is untouched
        self.lift_splat(batch)
        self.drop_feats(batch)
        if task.startswith("mlm"):
            return self.forward_mlm(batch, compute_loss)
        if task.startswith("sap"):
            return self.forward_sap(batch, compute_loss)
        if task.startswith("mrc"):
            return self.forward_mrc(batch, compute_loss)
        if task.startswith("og"):
            return self.forward_og(batch, compute_loss)
        if task.startswith("masksem"):
            return self.forward_masksem(batch, compute_loss)
        return self.forward_sem(batch, compute_loss)

    # -- tasks --------------------------------------------------------------------------------------
    def _host_kw(self, b):
        kw = {"view_lens_host": b.get("traj_vp_view_lens_cpu"), "obj_lens_host": b.get("traj_vp_obj_lens_cpu"),
              "gmap_csr": b.get("gmap_csr")}
        if getattr(self.config, "depth_feat_size", 0) > 0:           # continuous-environment fork
            kw["traj_view_dep_fts"] = b["traj_view_dep_fts"]
        return kw

    def forward_mrc(self, b, compute_loss):
        """pretrain_cmt.py:272-297: masked-region classification on the object tokens (KL to detector probs)."""
        _, _, obj_embeds, _ = self.bert(*self._cmt_args(b), return_gmap_embeds=False, **self._host_kw(b))
        sel = b["vp_obj_mrc_masks"]
        host = b.get("vp_obj_mrc_masks_cpu")
        if host is not None:
            pos = torch.nonzero(host.reshape(-1)).squeeze(1).to(sel.device, non_blocking=True)
            masked = obj_embeds.reshape(-1, obj_embeds.shape[-1]).index_select(0, pos)
            targets = b["vp_obj_probs"].reshape(-1, b["vp_obj_probs"].shape[-1]).index_select(0, pos)
        else:
            masked, targets = obj_embeds[sel], b["vp_obj_probs"][sel]
        pred = self.obj_classifier(masked).float()
        if compute_loss:
            loss = F.kl_div(F.log_softmax(pred, dim=-1), targets, reduction="none").sum(dim=1)
            return loss.mean() if compute_loss == "mean" else loss
        return pred, targets

    def forward_og(self, b, compute_loss):
        """pretrain_cmt.py:367-389: object grounding logits over the last viewpoint's objects."""
        _, _, obj_embeds, obj_masks = self.bert(*self._cmt_args(b), return_gmap_embeds=False, **self._host_kw(b))
        logits = self.og_head(obj_embeds).squeeze(2).float().masked_fill(obj_masks.logical_not(), -float("inf"))
        if compute_loss:
            loss = F.cross_entropy(logits, b["obj_labels"], reduction="none")
            return loss.mean() if compute_loss == "mean" else loss
        return logits

    def forward_mlm(self, b, compute_loss):
        txt_embeds = self.bert.forward_mlm(*self._cmt_args(b), **self._host_kw(b))
        labels = b["txt_labels"]
        st = b.get("_static")
        if st is not None:
            # loader-built positions of the masked tokens, padded to a fixed count (the padding re-reads row 0 and
            # carries zero weight): no nonzero() sync, no host->device copy inside the step, static GEMM shapes
            pos_in, n = st["mlm_pos"], st["mlm_n"]
            flat = txt_embeds.reshape(-1, txt_embeds.shape[-1])
            masked = ops.take_rows(flat, pos_in) if flat.is_cuda else flat.index_select(0, pos_in)
            if compute_loss == "mean":
                # cross-entropy straight from the head's logits (fp32 statistics, no fp32 copy of rows x vocabulary)
                per_row = ops.cross_entropy_rows(self.mlm_head(masked), st["mlm_targets"])
                return ops.weighted_mean(per_row, st["mlm_valid"], st["mlm_n_dev"])     # / row count of the batch in the buffers
            scores = self.mlm_head(masked).float()
            scores = scores[:n]
            if compute_loss:
                return F.cross_entropy(scores, st["mlm_targets"][:n], reduction="none")
            return scores
        host = b.get("txt_labels_cpu")
        if host is not None:        # host-known positions: no nonzero() sync (pretrain_cmt.py:254-256 has one)
            pos = torch.nonzero(host.reshape(-1) != -1).squeeze(1).to(labels.device, non_blocking=True)
        else:
            pos = torch.nonzero(labels.reshape(-1) != -1).squeeze(1)
        n = pos.numel()
        pad_to = _MLM_ROW_PAD
        if pad_to > 1 and n % pad_to:
            # opt-in (BEVBERT_MLM_ROW_PAD=128): round the data-dependent number of masked rows up so that the head's GEMM
            # problems repeat across batches (every new problem costs a hipBLASLt timing pass); the extra rows re-read
            # row 0 and are cut off again below, the rows that count are untouched
            pos_in = torch.cat([pos, pos.new_zeros(pad_to - n % pad_to)])
        else:
            pos_in = pos
        masked = txt_embeds.reshape(-1, txt_embeds.shape[-1]).index_select(0, pos_in)
        scores = self.mlm_head(masked).float()
        if pos_in is not pos:
            scores = scores[:n]
        if compute_loss:
            loss = F.cross_entropy(scores, labels.reshape(-1).index_select(0, pos), reduction="none")
            return loss.mean() if compute_loss == "mean" else loss
        return scores

    def forward_sap(self, b, compute_loss):
        cfg = self.config
        gmap_embeds, bev_embeds, _, _ = self.bert(*self._cmt_args(b), **self._host_kw(b))
        center = (cfg.bev_dim * cfg.bev_dim - 1) // 2
        G = gmap_embeds.shape[1]
        st = b.get("_static")
        if compute_loss and st is not None and G <= 64 and b["bev_cand_idxs"].shape[1] <= 62 and gmap_embeds.is_cuda:
            # static batch (loader-built fusion table on the device): the whole tail behind the heads -- masks, logit
            # fusion, three cross-entropies and their backward -- is one C-ABI launch each way (ops.sap_loss)
            cand_idxs = b["bev_cand_idxs"]
            graw = self.global_sap_head(gmap_embeds).squeeze(2)
            if "sap_cand_flat" in st:
                # loader-built flat row numbers of the candidate cells and of the centre cell: one row gather each, and ONE
                # zero-initialised gradient of the BEV states in backward (ops.take_rows) instead of an index_put, a slice
                # gradient and their sum over 28 224 rows
                Bc, Kc = cand_idxs.shape
                cand, cen = ops.take_rows(bev_embeds.reshape(-1, bev_embeds.shape[-1]), st["sap_cand_flat"], st["sap_center_flat"])
                cand = cand.view(Bc, Kc, -1)
            else:
                bi = torch.arange(cand_idxs.shape[0], device=cand_idxs.device)[:, None]
                cand, cen = bev_embeds[bi, cand_idxs], bev_embeds[:, center]
            lraw = self.local_sap_head(cand).squeeze(2)
            fraw = None if self.sap_fuse_linear is None else self.sap_fuse_linear(torch.cat([gmap_embeds[:, 0], cen], 1))
            loss = ops.sap_loss(graw, lraw, fraw, b["gmap_visited_masks"], b["gmap_lens"], b["bev_nav_masks"],
                                cand_idxs, st["sap_src"], st["sap_vis_c"], b["global_act_labels"],
                                b["local_act_labels"])
            return ops.weighted_mean(loss) if compute_loss == "mean" else loss
        if self.sap_fuse_linear is None:
            fuse_weights = 0.5
        else:
            fuse_weights = torch.sigmoid(self.sap_fuse_linear(
                torch.cat([gmap_embeds[:, 0], bev_embeds[:, center]], 1)).float())
        global_logits = self.global_sap_head(gmap_embeds).squeeze(2).float() * fuse_weights
        global_logits = global_logits.masked_fill(b["gmap_visited_masks"], -float("inf"))
        global_logits = global_logits.masked_fill(gen_seq_masks(b["gmap_lens"], G).logical_not(), -float("inf"))

        cand_idxs = b["bev_cand_idxs"]
        bi = torch.arange(cand_idxs.shape[0], device=cand_idxs.device)[:, None]
        cand_embeds = bev_embeds[bi, cand_idxs]
        cand_masks = b["bev_nav_masks"][bi, cand_idxs]
        local_logits = self.local_sap_head(cand_embeds).squeeze(2).float() * (1 - fuse_weights)
        local_logits = local_logits.masked_fill(cand_masks.logical_not(), -float("inf"))

        st = b.get("_static")
        if st is not None:          # loader-built fusion table (static_step.StaticBatch): nothing to copy inside the step
            src_d, vis_d = st["sap_src"], st["sap_vis_c"]
        else:
            cand_vpids = [[None] + c[-1] for c in b["traj_cand_vpids"]]
            src, vis_c = sap_fusion_indices(b["gmap_vpids"], _host_rows(b, "gmap_visited_masks"), cand_vpids, G,
                                            cand_idxs.shape[1])
            dev = global_logits.device
            src_d = torch.from_numpy(src).to(dev, non_blocking=True)
            vis_d = torch.from_numpy(vis_c).to(dev, non_blocking=True)
        fused_logits = fuse_sap_logits(global_logits, local_logits, src_d, vis_d)
        if compute_loss:
            loss = F.cross_entropy(global_logits, b["global_act_labels"], reduction="none") \
                + F.cross_entropy(local_logits, b["local_act_labels"], reduction="none") \
                + F.cross_entropy(fused_logits, b["global_act_labels"], reduction="none")
            return loss.mean() if compute_loss == "mean" else loss
        return global_logits, local_logits, fused_logits, b["global_act_labels"], b["local_act_labels"]

    def _sem_common(self, b, sel, compute_loss, sel2=None):
        bev_embeds = self.bert.forward_sem(*self._cmt_args(b), sem_pred_token=self.sem_pred_token, **self._host_kw(b))
        st = b.get("_static")
        if st is not None and compute_loss == "mean":
            # static row selection: the supervised cells (a data-dependent count that only the device knows: it depends
            # on which cells the splat filled) are compacted into a fixed number of rows >= the count (the loader knows
            # an upper bound: the number of masked cells); the padding re-reads row 0 and carries zero weight
            cap = st["sem_cap"]
            sems = b["bev_sems"].reshape(-1, b["bev_sems"].shape[-1])
            flat_e = bev_embeds.reshape(-1, bev_embeds.shape[-1])
            if flat_e.is_cuda and sems.dtype == torch.uint8:
                # one launch compacts the supervised cells (both masks) into the fixed-capacity index and leaves the weights and
                # the divisor; the loss reads the label rows through the index (ops.bce_rows) and is averaged by one launch
                idx, valid, denom = ops.sem_select(sel, sel2, cap, sems.shape[-1])
                masked = ops.take_rows(flat_e, idx)
                per_row = ops.bce_rows(self.local_sem_head(masked), sems.contiguous(), idx)
                return ops.weighted_mean(per_row, valid, denom)
            flat = (sel if sel2 is None else sel & sel2).reshape(-1)
            idx = torch.nonzero_static(flat, size=cap, fill_value=0).squeeze(1)
            count = flat.sum()
            valid = (torch.arange(cap, device=flat.device) < count).to(torch.float32)
            masked = flat_e.index_select(0, idx)
            sem_logits = self.local_sem_head(masked).float()
            sem_labels = sems.index_select(0, idx).float()
            per = F.binary_cross_entropy_with_logits(sem_logits, sem_labels, reduction="none")
            return (per * valid[:, None]).sum() / (count.to(torch.float32) * per.shape[1])
        if sel2 is not None:
            sel = sel & sel2
        masked = bev_embeds[sel]                               # data-dependent row count: one sync, as the reference
        sem_logits = self.local_sem_head(masked).float()
        sem_labels = b["bev_sems"][sel].float()
        if compute_loss:
            loss = F.binary_cross_entropy_with_logits(sem_logits, sem_labels, reduction="none")
            return loss.mean() if compute_loss == "mean" else loss
        return sem_logits, sem_labels

    def forward_sem(self, b, compute_loss):
        return self._sem_common(b, b["bev_sem_masks"], compute_loss)

    def forward_masksem(self, b, compute_loss):
        mrc = b["bev_mrc_masks"]
        b["bev_fts"] = b["bev_fts"].detach().masked_fill(mrc.unsqueeze(-1), 0)        # pretrain_cmt.py:423-424
        return self._sem_common(b, b["bev_sem_masks"], compute_loss, sel2=mrc)
