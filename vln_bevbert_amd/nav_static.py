"""Captured navigation steps for inference rollouts (validation / test: map_nav_src/r2r/agent.py:436-560 under
``torch.no_grad()``).

Per navigation step the reference calls the model twice -- ``vln_bert('panorama', ...)`` and ``vln_bert('navigation',
...)`` -- and each call is ~200 small launches for a batch of 32, so the rollout is bound by the host's launch rate,
not by the GPU (round 3, ``scripts/gpu_profile_nav.sh``: 3.9 ms of kernel time in a 6.7 ms step).  ``NavGraphRunner``
keeps, per input-shape bucket, a set of static device buffers and ONE hipGraph of the device half of each mode:

* panorama: the shapes are fixed by (batch, views); one graph;
* navigation: the global map grows by the step, so its node axis is padded to a multiple of ``g_step`` (padding nodes
  are masked out exactly like the in-batch padding the agent already produces: ``gmap_masks`` False, zero embeddings,
  zero distances) and the candidate axis to a multiple of ``c_step`` (padding candidates are never gathered by the
  logit fusion); a 15-step episode needs two or three graphs.

What stays on the host is what is host data anyway: the viewpoint-id lists of the observations and the fusion index
table built from them (``sap_fusion_indices``); they are copied into the static buffers with the rest of the step's
inputs.  A bucket is run eagerly twice (library GEMM plans are chosen by timing on first use) and captured on its third
use.  Outputs are views of static buffers: they are valid until the next call of the same mode.
"""
import numpy as np
import torch

from .graph_map import HostFeed
from .pretrain_cmt import fuse_sap_logits, sap_fusion_indices


def _pad_to(n, step):
    return max(step, (n + step - 1) // step * step)


class _Bucket:
    def __init__(self):
        self.inputs, self.outputs, self.graph, self.uses = {}, None, None, 0
        self.filled = {}          # name -> shape of the last (smaller than the buffer) tensor written into the buffer


class NavGraphRunner:
    def __init__(self, model, g_step=8, c_step=8, eager_uses=2):
        """``model``: nav_model.VLNBert in eval mode, parameters already in the arena."""
        self.model, self.net = model, model.vln_bert
        self.g_step, self.c_step, self.eager_uses = g_step, c_step, eager_uses
        self.buckets = {}
        self.stats = {"replays": 0, "eager": 0, "captures": 0}
        self.graph_error = None
        self.pool = None
        self.feed = None
        import os
        # 1 = the K|V projections of the instruction are computed once per episode instead of inside every captured step.
        # Measured (profiles/r04ae_nav_text_kv_cache_ab.txt): 3.23 / 3.58 ms per navigation step with, 3.21 / 3.53 without
        # (no feedback / feedback) -- the two GEMMs hide behind the panorama branch.  Off.
        self.text_cache = os.environ.get("BEVBERT_NAV_TEXT_CACHE", "0") == "1"
        self.shared = {}            # episode-constant static inputs (instruction states, masks, their K|V projections)
        self._text_src = None       # weak reference + version of the txt_embeds tensor the shared inputs were built from

    # ------------------------------------------------------------------------------------------------ plumbing
    def _run(self, key, feeds, fn, shapes=None, static=None):
        """Copy ``feeds`` (name -> tensor) into the bucket's static inputs -- ``shapes`` names the (padded) buffer shape
        of the tensors that are smaller than their buffer; the padding is zero -- run ``fn(static inputs)`` eagerly or
        replay its graph, return its (static) outputs.  ``static``: name -> buffer the caller keeps up to date itself
        (shared by every bucket: the instruction's tensors change once per episode, not per step)."""
        assert not self.model.training, "NavGraphRunner serves inference rollouts (model.eval())"
        shapes = shapes or {}
        b = self.buckets.get(key)
        if b is None:
            b = self.buckets[key] = _Bucket()
            b.inputs = {k: torch.zeros(shapes.get(k, v.shape), dtype=v.dtype, device=v.device) for k, v in feeds.items()}
            b.inputs.update(static or {})         # buffers that are filled elsewhere (once per episode), not per call
        for k, v in feeds.items():
            buf = b.inputs[k]
            if v.shape == buf.shape:
                buf.copy_(v, non_blocking=True)
                b.filled[k] = tuple(v.shape)      # a later, smaller fill must clear what this one wrote
                continue
            last = b.filled.get(k)
            if last is not None and any(l > n for l, n in zip(last, v.shape)):
                buf.zero_()                       # the map shrank (a new episode): clear what the last fill left behind
            buf[tuple(slice(0, n) for n in v.shape)].copy_(v, non_blocking=True)
            b.filled[k] = tuple(v.shape)
        b.uses += 1
        if b.graph is not None:
            b.graph.replay()
            self.stats["replays"] += 1
            return b.outputs
        if b.uses <= self.eager_uses or self.graph_error is not None:
            self.stats["eager"] += 1
            with torch.no_grad():
                return fn(b.inputs)
        try:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                out = fn(b.inputs)
            self.pool = self.pool or g.pool()
            b.graph, b.outputs = g, out
            self.stats["captures"] += 1
            g.replay()
            self.stats["replays"] += 1
            return out
        except Exception as e:      # noqa: BLE001 -- a refused capture is reported, the rollout goes on eagerly
            self.graph_error = f"{type(e).__name__}: {e}"
            from . import lib
            lib.load().bevbert_hip_error_reset()
            torch.cuda.synchronize()
            self.stats["eager"] += 1
            with torch.no_grad():
                return fn(b.inputs)

    # ------------------------------------------------------------------------------------------------ panorama
    def panorama(self, batch):
        """``model('panorama', batch)`` for batches without object tokens; returns (pano_embeds, pano_masks)."""
        if batch.get("obj_img_fts") is not None:
            with torch.no_grad():
                return self.model("panorama", batch)
        feeds = {k: batch[k] for k in ("view_img_fts", "loc_fts", "nav_types", "view_lens")}
        key = ("panorama",) + tuple(feeds["view_img_fts"].shape)

        def fn(x):
            return self.net.forward_panorama_per_step(x["view_img_fts"], None, x["loc_fts"], x["nav_types"],
                                                      x["view_lens"], None)
        return self._run(key, feeds, fn)

    # ------------------------------------------------------------------------------------------------ navigation
    def navigation(self, nav):
        """``model('navigation', nav)`` (no object tokens).  ``nav`` is the dict the agent builds: the outputs of
        ``GraphMapBatch.nav_gmap_variable`` / ``bev_inputs`` plus txt_embeds, txt_masks, bev_fts, bev_pos_fts."""
        if nav.get("obj_embeds") is not None:
            with torch.no_grad():
                return self.model("navigation", nav)
        B, G = nav["gmap_masks"].shape
        C = nav["bev_cand_idxs"].shape[1]
        Gp, Cp = _pad_to(G, self.g_step), _pad_to(C, self.c_step)
        vis_host = nav.get("gmap_visited_masks_cpu")
        vis_host = (vis_host if vis_host is not None else nav["gmap_visited_masks"].cpu()).tolist()
        src, vis_c = sap_fusion_indices(nav["gmap_vpids"], vis_host, nav["bev_cand_vpids"], Gp, Cp)
        dev = nav["gmap_masks"].device
        if self.feed is None:
            self.feed = HostFeed.shared(dev)
        idx = self.feed({"src": src, "vis_c": vis_c})
        names = ("gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_masks",
                 "gmap_pair_dists", "gmap_visited_masks", "bev_fts", "bev_pos_fts", "bev_nav_masks", "bev_cand_idxs")
        feeds = {k: nav[k] for k in names}
        static = self._text_inputs(nav["txt_embeds"], nav["txt_masks"])
        feeds.update(src=idx["src"], vis_c=idx["vis_c"])
        shapes = {"gmap_img_embeds": (B, Gp, nav["gmap_img_embeds"].shape[2]), "gmap_step_ids": (B, Gp),
                  "gmap_pos_fts": (B, Gp, nav["gmap_pos_fts"].shape[2]), "gmap_masks": (B, Gp),
                  "gmap_pair_dists": (B, Gp, Gp), "gmap_visited_masks": (B, Gp), "bev_cand_idxs": (B, Cp)}
        key = ("navigation", B, Gp, Cp, nav["txt_embeds"].shape[1], nav["bev_fts"].shape[1])
        out = self._run(key, feeds, self._navigation_device, shapes, static)
        return {"gmap_embeds": out["gmap_embeds"][:, :G], "global_logits": out["global_logits"][:, :G],
                "local_logits": out["local_logits"][:, :C], "fused_logits": out["fused_logits"][:, :G],
                "obj_logits": None}

    def _text_inputs(self, txt_embeds, txt_masks):
        """Static buffers of the instruction: its states and mask, and their K|V projections for every cross-attention
        layer of the two map encoders (vilmodel.CrossmodalEncoder.packed_kv) -- written when a NEW txt_embeds tensor
        arrives (the agent encodes the instruction once per episode, map_nav_src/r2r/agent.py:426-431, and hands the same
        tensor to every navigation step), shared by all shape buckets; the captured steps read them in place."""
        import weakref
        key = (tuple(txt_embeds.shape), txt_embeds.dtype, txt_embeds.device)
        bufs = self.shared.get(key)
        src = self._text_src
        if bufs is not None and src is not None and src[0]() is txt_embeds and src[1] == txt_embeds._version \
                and src[2]() is txt_masks and src[3] == txt_masks._version and src[4] == key:
            return bufs
        net = self.net
        with torch.no_grad():
            kv = {"g_kv": net.global_encoder.encoder.packed_kv(txt_embeds) if self.text_cache else None,
                  "b_kv": net.local_encoder.encoder.packed_kv(txt_embeds) if self.text_cache else None}
            if bufs is None:
                bufs = self.shared[key] = {"txt_embeds": torch.empty_like(txt_embeds), "txt_masks": torch.empty_like(txt_masks)}
                bufs.update({k: torch.empty_like(v) for k, v in kv.items() if v is not None})
            bufs["txt_embeds"].copy_(txt_embeds)
            bufs["txt_masks"].copy_(txt_masks)
            for k, v in kv.items():
                if v is not None:
                    bufs[k].copy_(v)
        self._text_src = (weakref.ref(txt_embeds), txt_embeds._version, weakref.ref(txt_masks), txt_masks._version, key)
        return bufs

    def _navigation_device(self, x):
        """The device half of GlocalTextPathNavCMT.forward_navigation_per_step (nav_model.py) on static inputs."""
        net = self.net
        cd = x["txt_embeds"].dtype
        g_in = net.global_encoder.pos_step_embedding(x["gmap_img_embeds"].to(cd), x["gmap_step_ids"], x["gmap_pos_fts"])
        g_kvs = net.global_encoder.encoder.split_kv(x["g_kv"]) if "g_kv" in x else None      # cached per episode
        b_kvs = net.local_encoder.encoder.split_kv(x["b_kv"]) if "b_kv" in x else None
        gmap_embeds = net.global_encoder(x["txt_embeds"], x["txt_masks"], g_in, x["gmap_masks"], x["gmap_pair_dists"],
                                         txt_kvs=g_kvs)
        bev_embeds, _ = net.local_encoder(x["txt_embeds"], x["txt_masks"], x["bev_fts"], x["bev_pos_fts"], None,
                                          x["bev_nav_masks"], None, None, txt_kvs=b_kvs)
        if net.sap_fuse_linear is None:
            fuse_weights = 0.5
        else:
            center = (net.bev_dim * net.bev_dim - 1) // 2
            fuse_weights = torch.sigmoid(net.sap_fuse_linear(
                torch.cat([gmap_embeds[:, 0], bev_embeds[:, center]], 1)).float())
        global_logits = net.global_sap_head(gmap_embeds).squeeze(2).float() * fuse_weights
        global_logits = global_logits.masked_fill(x["gmap_visited_masks"], -float("inf"))
        global_logits = global_logits.masked_fill(x["gmap_masks"].logical_not(), -float("inf"))
        bi = torch.arange(x["bev_cand_idxs"].shape[0], device=global_logits.device)[:, None]
        cand_embeds = bev_embeds[bi, x["bev_cand_idxs"]]
        cand_masks = x["bev_nav_masks"][bi, x["bev_cand_idxs"]]
        local_logits = net.local_sap_head(cand_embeds).squeeze(2).float() * (1 - fuse_weights)
        local_logits = local_logits.masked_fill(cand_masks.logical_not(), -float("inf"))
        fused_logits = fuse_sap_logits(global_logits, local_logits, x["src"], x["vis_c"])
        return {"gmap_embeds": gmap_embeds, "global_logits": global_logits, "local_logits": local_logits,
                "fused_logits": fused_logits}

    def captured_graphs(self):
        return sum(b.graph is not None for b in self.buckets.values())
