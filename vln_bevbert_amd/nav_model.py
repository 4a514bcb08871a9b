"""Host-side mirror of the fine-tuning model: map_nav_src/models/vilmodel.py (GlocalTextPathNavCMT) and
map_nav_src/models/model.py (VLNBert).  Three per-step modes over the same fused blocks as pre-training."""
import torch
from torch import nn

from . import ops
from .config import BevBertConfig
from .pretrain_cmt import ClsPrediction, fuse_sap_logits, sap_fusion_indices
from .vilmodel import (BertEmbeddings, GlobalMapEncoder, ImageEmbeddings, LanguageEncoder, LocalBEVEncoder,
                       _all_ones_to_none, ensure_arena, finalize, from_pretrained)


class GlocalTextPathNavCMT(nn.Module):
    def __init__(self, config):
        super().__init__()
        config = BevBertConfig.adopt(config)     # map_nav_src/models/vlnbert_init.py:50-76 builds a PretrainedConfig
        self.config = config
        self.bev_dim = config.bev_dim
        self.embeddings = BertEmbeddings(config)
        self.lang_encoder = LanguageEncoder(config)
        self.img_embeddings = ImageEmbeddings(config)
        self.local_encoder = LocalBEVEncoder(config)
        self.global_encoder = GlobalMapEncoder(config)
        self.global_sap_head = ClsPrediction(config.hidden_size)
        self.local_sap_head = ClsPrediction(config.hidden_size)
        self.sap_fuse_linear = ClsPrediction(config.hidden_size, input_size=config.hidden_size * 2) \
            if config.glocal_fuse else None
        if config.obj_feat_size > 0:
            self.og_head = ClsPrediction(config.hidden_size)
        self.init_weights()
        if config.fix_lang_embedding or config.fix_local_branch:          # vilmodel.py:727-741
            for m in (self.embeddings, self.lang_encoder):
                for p in m.parameters():
                    p.requires_grad = False
        if config.fix_pano_embedding or config.fix_local_branch:
            for p in self.img_embeddings.parameters():
                p.requires_grad = False
        if config.fix_local_branch:
            for m in (self.local_encoder, self.local_sap_head) + ((self.og_head,) if hasattr(self, "og_head") else ()):
                for p in m.parameters():
                    p.requires_grad = False

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                m.weight.data.normal_(mean=0.0, std=0.02)
                if isinstance(m, nn.Linear) and m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.LayerNorm):
                m.weight.data.fill_(1.0)
                m.bias.data.zero_()

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, config=None, state_dict=None, **kwargs):
        """The call map_nav_src/models/vlnbert_init.py:78-81 makes; see vilmodel.from_pretrained."""
        return from_pretrained(cls, pretrained_model_name_or_path, config, state_dict)

    def finalize(self, device, compute_dtype=torch.float32, residual=None):
        return finalize(self, device, compute_dtype, residual)

    def forward_text(self, txt_ids, txt_masks):
        return self.lang_encoder(self.embeddings(txt_ids), txt_masks)

    def forward_panorama_per_step(self, view_img_fts, obj_img_fts, loc_fts, nav_types, view_lens, obj_lens):
        return self.img_embeddings.embed(view_img_fts, loc_fts, nav_types, view_lens,
                                         self.embeddings.token_type_embeddings, obj_img_fts, obj_lens)

    def forward_navigation_per_step(self, txt_embeds, txt_masks, gmap_img_embeds, gmap_step_ids, gmap_pos_fts,
                                    gmap_masks, gmap_pair_dists, gmap_visited_masks, gmap_vpids,
                                    bev_fts, bev_pos_fts, bev_masks, bev_nav_masks, bev_cand_idxs, bev_cand_vpids,
                                    obj_embeds, obj_masks, gmap_visited_masks_host=None):
        cd = txt_embeds.dtype
        g_in = self.global_encoder.pos_step_embedding(gmap_img_embeds.to(cd), gmap_step_ids, gmap_pos_fts)
        gmap_embeds = self.global_encoder(txt_embeds, txt_masks, g_in, gmap_masks, gmap_pair_dists)
        bev_embeds, obj_embeds = self.local_encoder(txt_embeds, txt_masks, bev_fts, bev_pos_fts,
                                                    _all_ones_to_none(bev_masks), bev_nav_masks,
                                                    None if obj_embeds is None else obj_embeds.to(cd), obj_masks)
        if self.sap_fuse_linear is None:
            fuse_weights = 0.5
        else:
            center = (self.bev_dim * self.bev_dim - 1) // 2
            fuse_weights = torch.sigmoid(self.sap_fuse_linear(
                torch.cat([gmap_embeds[:, 0], bev_embeds[:, center]], 1)).float())
        global_logits = self.global_sap_head(gmap_embeds).squeeze(2).float() * fuse_weights
        global_logits = global_logits.masked_fill(gmap_visited_masks, -float("inf"))
        global_logits = global_logits.masked_fill(gmap_masks.logical_not(), -float("inf"))
        bi = torch.arange(bev_cand_idxs.shape[0], device=bev_cand_idxs.device)[:, None]
        cand_embeds = bev_embeds[bi, bev_cand_idxs]
        cand_masks = bev_nav_masks[bi, bev_cand_idxs]
        local_logits = self.local_sap_head(cand_embeds).squeeze(2).float() * (1 - fuse_weights)
        local_logits = local_logits.masked_fill(cand_masks.logical_not(), -float("inf"))
        vis_host = gmap_visited_masks_host if gmap_visited_masks_host is not None else gmap_visited_masks.tolist()
        src, vis_c = sap_fusion_indices(gmap_vpids, vis_host, bev_cand_vpids, gmap_embeds.shape[1],
                                        bev_cand_idxs.shape[1])
        dev = global_logits.device
        fused_logits = fuse_sap_logits(global_logits, local_logits, torch.from_numpy(src).to(dev, non_blocking=True),
                                       torch.from_numpy(vis_c).to(dev, non_blocking=True))
        obj_logits = None
        if obj_embeds is not None:                                      # map_nav_src/models/vilmodel.py:873-877
            obj_logits = self.og_head(obj_embeds).squeeze(2).float().masked_fill(obj_masks.logical_not(), -float("inf"))
        return {"gmap_embeds": gmap_embeds, "global_logits": global_logits, "local_logits": local_logits,
                "fused_logits": fused_logits, "obj_logits": obj_logits}

    def forward(self, mode, batch, **kwargs):
        ensure_arena(self)
        if mode == "language":
            return self.forward_text(batch["txt_ids"], batch["txt_masks"])
        if mode == "panorama":
            return self.forward_panorama_per_step(batch["view_img_fts"], batch.get("obj_img_fts"), batch["loc_fts"],
                                                  batch["nav_types"], batch["view_lens"], batch.get("obj_lens"))
        if mode == "navigation":
            return self.forward_navigation_per_step(
                batch["txt_embeds"], batch["txt_masks"], batch["gmap_img_embeds"], batch["gmap_step_ids"],
                batch["gmap_pos_fts"], batch["gmap_masks"], batch["gmap_pair_dists"], batch["gmap_visited_masks"],
                batch["gmap_vpids"], batch["bev_fts"], batch["bev_pos_fts"], batch["bev_masks"],
                batch["bev_nav_masks"], batch["bev_cand_idxs"], batch["bev_cand_vpids"],
                batch.get("obj_embeds"), batch.get("obj_masks"),
                gmap_visited_masks_host=batch.get("gmap_visited_masks_cpu"))
        raise NotImplementedError("wrong mode: %s" % mode)


class VLNBert(nn.Module):
    """map_nav_src/models/model.py:12-41: dropout on the text / panorama embeddings around the three modes."""

    def __init__(self, config, feat_dropout=0.4):
        super().__init__()
        self.vln_bert = GlocalTextPathNavCMT(config)
        self.drop_env = nn.Dropout(feat_dropout)                  # model.py:19

    feat_dropout = property(lambda self: self.drop_env.p, lambda self, v: setattr(self.drop_env, "p", v))

    def forward(self, mode, batch):
        batch = dict(batch)
        if mode == "language":
            return self.vln_bert(mode, batch)
        if mode == "panorama":
            batch["view_img_fts"] = ops.dropout(batch["view_img_fts"], self.feat_dropout, self.training)
            if batch.get("obj_img_fts") is not None:
                batch["obj_img_fts"] = ops.dropout(batch["obj_img_fts"], self.feat_dropout, self.training)
            return self.vln_bert(mode, batch)
        if mode == "navigation":
            batch["bev_fts"] = ops.dropout(batch["bev_fts"], self.feat_dropout, self.training)   # model.py:36
            return self.vln_bert(mode, batch)
        raise NotImplementedError("wrong mode: %s" % mode)


def remap_pretrain_checkpoint(state_dict, model=None):
    """map_nav_src/models/vlnbert_init.py:39-46: pre-training keys -> fine-tuning module keys
    ('module.' stripped; '*_head' / 'sap_fuse' keys gain the 'bert.' prefix; then 'bert.' is the model root).

    With ``model`` (a GlocalTextPathNavCMT) the mapped dict is filtered to the model's own keys so that it loads with
    ``strict=True``, and the function returns ``(mapped, missing, unexpected)`` -- what the reference's
    ``from_pretrained`` reports: keys the model has but the checkpoint lacks, and checkpoint keys the fine-tuning
    model has no use for (the pre-training-only heads: mlm_head.*, local_sem_head.*, obj_classifier.*, ...)."""
    out = {}
    for k, v in state_dict.items():
        if k.startswith("module."):
            k = k[7:]
        if "_head" in k or "sap_fuse" in k:
            k = "bert." + k
        if k.startswith("bert."):
            out[k[5:]] = v
    if model is None:
        return out
    own = model.state_dict()
    unexpected = sorted(k for k in out if k not in own)
    missing = sorted(k for k in own if k not in out)
    return {k: v for k, v in out.items() if k in own}, missing, unexpected
