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


# ----------------------------------------------------------------------------------------------------------------------
# Captured TRAINING rollouts (map_nav_src/r2r/agent_base.py:174-217 train(); r2r/agent.py:436-600 rollout with
# train_ml: 15 forwards, ONE backward through all of them, clip, optimiser step).
#
# The device half of every model call of an episode is a SEGMENT: its inputs are static buffers (the differentiable ones --
# the instruction states, the map's node embeddings -- are leaf tensors that require grad), its forward and its backward
# are one hipGraph each.  An autograd Function bridges the segments and the eager glue between them (map updates, the
# imitation loss): forward = copy the inputs in, replay the forward graph; backward = copy the output gradients in, replay
# the backward graph (which also issues the deferred weight-gradient / reduction work of that segment: ParamArena.sync
# inside the capture), hand the input gradients back to autograd.  A segment belongs to ONE step index of the episode: its
# saved activations live in the graph's pool from its forward replay to its backward replay, so step t and step t + 1 may
# not share a graph even when their shapes agree.
#
# Life of a runner: ``eager_uses`` episodes run the same segment code eagerly (the library's GEMM plans settle: they are
# chosen by timing on first use); one CAPTURE episode follows -- the forward graphs are captured in rollout order (and
# replayed, the glue needs their outputs), then the backward graphs in reverse order with placeholder gradients, all in
# one memory pool: the allocator may hand a block that one capture freed to a later one, which is only sound if the graphs
# are later replayed in the order they were captured in -- forward 1..N, backward N..1 is exactly the order of a rollout.
# The capture episode does not train (nothing of its backward executes).  From then on episodes whose segments all have
# graphs replay them; a segment without one (a new shape bucket) runs eagerly.
class _TrainSeg:
    def __init__(self, key):
        self.key, self.inputs, self.diff, self.fn = key, {}, (), None
        self.outs, self.out_names, self.grad_outs, self.grad_ins = None, (), None, None
        self.fwd, self.bwd, self.uses, self.filled = None, None, 0, {}


class _GradTap(torch.autograd.Function):
    """Identity on a differentiable static input of a segment; the backward keeps the incoming gradient for the bridge
    (``seg.grad_ins``) and ends the graph there.  Letting the gradient reach the leaf instead would run its AccumulateGrad
    node on the stream the buffer was created on -- outside the capture: the engine then synchronises with the legacy default
    stream in the middle of a capture (round 6: the process died in the first captured backward)."""

    @staticmethod
    def forward(ctx, x, seg, name):
        ctx.seg, ctx.name = seg, name
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.seg.grad_ins[ctx.name] = g
        return None, None, None


class _SegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, seg, anchor, *diff_tensors):
        # ``anchor``: a persistent scalar that requires grad -- a segment without differentiable inputs (the panorama encoder
        # reads features only) must still be a node of the autograd graph, or its parameter gradients would never be issued
        with torch.no_grad():
            for name, x in zip(seg.diff, diff_tensors):
                runner._fill(seg, name, x)
        if seg.fwd is not None:
            seg.fwd.replay()
            runner.stats["replays"] += 1
        else:
            runner._forward_eager(seg)
            runner.stats["eager"] += 1
        ctx.runner, ctx.seg = runner, seg
        ctx.in_shapes = [tuple(x.shape) for x in diff_tensors]
        outs = tuple(seg.outs[n].detach() for n in seg.out_names)
        ctx.mark_non_differentiable(*[o for n, o in zip(seg.out_names, outs) if not seg.outs[n].requires_grad])
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        runner, seg = ctx.runner, ctx.seg
        gin = runner._backward(seg, dict(zip(seg.out_names, grads)))
        res = []
        for name, shp in zip(seg.diff, ctx.in_shapes):
            g = gin.get(name)
            res.append(None if g is None else g[tuple(slice(0, n) for n in shp)])
        return (None, None, None) + tuple(res)


class NavTrainRunner(NavGraphRunner):
    """``panorama`` / ``navigation`` of a TRAINING rollout (see the block comment above).  ``begin_episode()`` before each
    rollout; ``end_capture()`` after the forward of the capture episode (``capturing`` says when that is)."""

    def __init__(self, model, arena, g_step=8, c_step=8, eager_uses=2, graphs=True):
        super().__init__(model, g_step=g_step, c_step=c_step, eager_uses=eager_uses)
        self.arena, self.graphs = arena, graphs
        self.segs, self.episode, self.capturing = {}, 0, False
        self._order, self._t = [], {"panorama": 0, "navigation": 0}
        self._pool = None
        self._anchor = None

    # ---- episode protocol
    def begin_episode(self):
        assert self.model.training, "NavTrainRunner serves training rollouts (model.train())"
        self._t = {"panorama": 0, "navigation": 0}
        self._order = []
        self.capturing = (self.graphs and self.graph_error is None and self.episode == self.eager_uses
                          and torch.cuda.is_available())
        self.episode += 1

    def end_capture(self):
        """After the forward pass of the capture episode: capture the backward graphs, last segment first."""
        if not self.capturing:
            return
        try:
            for seg in reversed(self._order):
                self._capture_backward(seg)
        except Exception as e:      # noqa: BLE001 -- a refused capture is reported, training goes on eagerly
            self._capture_failed(e)
        self.capturing = False
        self._order = []

    def _capture_failed(self, e):
        import traceback
        self.graph_error = f"{type(e).__name__}: {e}"[:400]
        self.graph_traceback = "".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:]
        from . import lib, ops
        if torch.cuda.is_current_stream_capturing():
            ops.join_captured_side_streams()
        lib.load().bevbert_hip_error_reset()
        torch.cuda.synchronize()
        ops.WgradStream.drop_pending()
        ops.ReduceQueue.drop_pending()
        for seg in self.segs.values():
            seg.fwd = seg.bwd = None

    # ---- plumbing
    def _fill(self, seg, name, v):
        buf = seg.inputs[name]
        if v.shape == buf.shape:
            buf.copy_(v, non_blocking=True)
            seg.filled[name] = tuple(v.shape)
            return
        last = seg.filled.get(name)
        if last is not None and any(l > n for l, n in zip(last, v.shape)):
            buf.zero_()
        buf[tuple(slice(0, n) for n in v.shape)].copy_(v, non_blocking=True)
        seg.filled[name] = tuple(v.shape)

    @staticmethod
    def _tapped(seg):
        seg.grad_ins = {}
        return {k: (_GradTap.apply(v, seg, k) if k in seg.diff else v) for k, v in seg.inputs.items()}

    def _forward_eager(self, seg):
        with torch.enable_grad():
            seg.outs = seg.fn(self._tapped(seg))

    def _backward(self, seg, grads):
        req = [n for n in seg.out_names if seg.outs[n].requires_grad]
        if seg.bwd is not None:
            from . import lib
            for n in req:
                g = grads.get(n)
                if g is None:
                    lib.call("bevbert_zero", seg.grad_outs[n].data_ptr(), seg.grad_outs[n].numel() * seg.grad_outs[n].element_size(),
                             lib.stream())
                else:
                    seg.grad_outs[n].copy_(g, non_blocking=True)
            seg.bwd.replay()
            self.stats["replays"] += 1
            return seg.grad_ins
        outs, gs = [], []
        for n in req:
            g = grads.get(n)
            outs.append(seg.outs[n])
            gs.append(torch.zeros_like(seg.outs[n]) if g is None else g.contiguous())
        torch.autograd.backward(outs, gs)
        self.arena.sync()                      # the segment's deferred weight-gradient / reduction work goes out HERE in
        return seg.grad_ins                    # both modes: the arena sees the same additions in the same order

    def _captured(self, body):
        """Run ``body`` inside a stream capture; an exception inside it is carried OUT of the capture (with every forked side
        stream joined first: a capture with unjoined forks cannot end, and the stream would stay in capture mode)."""
        from . import ops
        torch.cuda.synchronize()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        g = torch.cuda.CUDAGraph()
        err = None
        try:
            with torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
                try:
                    body()
                except Exception as e:      # noqa: BLE001
                    err = e
                    ops.join_captured_side_streams()
        except Exception as e:      # noqa: BLE001 -- ending the capture failed as well
            err = err or e
        if err is not None:
            raise err
        self.stats["captures"] += 1
        return g

    def _capture_forward(self, seg):
        def body():
            with torch.enable_grad():
                seg.outs = seg.fn(self._tapped(seg))
        seg.fwd = self._captured(body)
        seg.fwd.replay()
        self._order.append(seg)

    def _capture_backward(self, seg):
        req = [n for n in seg.out_names if seg.outs[n].requires_grad]
        seg.grad_outs = {n: torch.zeros_like(seg.outs[n]) for n in req}

        def body():
            torch.autograd.backward([seg.outs[n] for n in req], [seg.grad_outs[n] for n in req])
            self.arena.sync()
        seg.bwd = self._captured(body)          # (its _GradTap nodes left the input gradients in seg.grad_ins: tensors of the
                                                # graph's pool, rewritten by every replay)

    def _segment(self, mode, key, feeds, diff, fn, shapes, out_names):
        t = self._t[mode]
        self._t[mode] += 1
        key = (mode, t) + tuple(key)
        seg = self.segs.get(key)
        if seg is None:
            seg = self.segs[key] = _TrainSeg(key)
            seg.diff, seg.fn, seg.out_names = tuple(diff), fn, tuple(out_names)
            for k, v in feeds.items():
                buf = torch.zeros(shapes.get(k, v.shape), dtype=v.dtype, device=v.device)
                seg.inputs[k] = buf.requires_grad_(True) if k in diff else buf
        seg.uses += 1
        with torch.no_grad():
            for k, v in feeds.items():
                if k not in seg.diff:
                    self._fill(seg, k, v)
        if self.capturing and seg.fwd is None:
            try:
                with torch.no_grad():
                    for k in seg.diff:
                        self._fill(seg, k, feeds[k])
                self._capture_forward(seg)
                # (leaves that require grad where the real outputs do: the glue behind them -- the map's functional update
                # path, the choice of the next segments' differentiable inputs -- must see what a training episode sees)
                return {n: seg.outs[n].detach().requires_grad_(seg.outs[n].requires_grad) for n in seg.out_names}
            except Exception as e:      # noqa: BLE001
                self._capture_failed(e)
        if self._anchor is None:
            self._anchor = torch.zeros((), device=next(iter(feeds.values())).device, requires_grad=True)
        outs = _SegFn.apply(self, seg, self._anchor, *[feeds[k] for k in seg.diff])
        return dict(zip(seg.out_names, outs))

    # ---- the two modes
    def panorama(self, batch):
        if batch.get("obj_img_fts") is not None:
            return self.model("panorama", batch)
        feeds = {k: batch[k] for k in ("view_img_fts", "loc_fts", "nav_types", "view_lens")}
        model = self.model

        def fn(x):
            pe, pm = model("panorama", {"view_img_fts": x["view_img_fts"], "obj_img_fts": None, "loc_fts": x["loc_fts"],
                                        "nav_types": x["nav_types"], "view_lens": x["view_lens"], "obj_lens": None})
            return {"pano_embeds": pe, "pano_masks": pm}
        out = self._segment("panorama", tuple(feeds["view_img_fts"].shape), feeds, (), fn, {}, ("pano_embeds", "pano_masks"))
        return out["pano_embeds"], out["pano_masks"]

    def navigation(self, nav):
        if nav.get("obj_embeds") is not None:
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
        names = ("txt_embeds", "txt_masks", "gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_masks",
                 "gmap_pair_dists", "gmap_visited_masks", "bev_fts", "bev_pos_fts", "bev_nav_masks", "bev_cand_idxs")
        feeds = {k: nav[k] for k in names}
        feeds["gmap_img_embeds"] = feeds["gmap_img_embeds"].to(nav["txt_embeds"].dtype)
        feeds.update(src=idx["src"], vis_c=idx["vis_c"])
        shapes = {"gmap_img_embeds": (B, Gp, nav["gmap_img_embeds"].shape[2]), "gmap_step_ids": (B, Gp),
                  "gmap_pos_fts": (B, Gp, nav["gmap_pos_fts"].shape[2]), "gmap_masks": (B, Gp),
                  "gmap_pair_dists": (B, Gp, Gp), "gmap_visited_masks": (B, Gp), "bev_cand_idxs": (B, Cp)}
        diff = tuple(k for k in ("txt_embeds", "gmap_img_embeds") if feeds[k].requires_grad)
        drop = self.model

        def fn(x):
            y = dict(x)
            from . import ops
            y["bev_fts"] = ops.dropout(x["bev_fts"], drop.feat_dropout, True)        # VLNBert.forward('navigation'), model.py:36
            return self._navigation_device(y)
        key = (B, Gp, Cp, nav["txt_embeds"].shape[1], nav["bev_fts"].shape[1], diff)
        out = self._segment("navigation", key, feeds, diff, fn, shapes,
                            ("gmap_embeds", "global_logits", "local_logits", "fused_logits"))
        return {"gmap_embeds": out["gmap_embeds"][:, :G], "global_logits": out["global_logits"][:, :G],
                "local_logits": out["local_logits"][:, :C], "fused_logits": out["fused_logits"][:, :G], "obj_logits": None}

    def captured_graphs(self):
        return sum((s.fwd is not None) + (s.bwd is not None) for s in self.segs.values())
