#!/usr/bin/env python3
"""Fine-tune rollout timing (BASELINE.json configs[4]: R2R fine-tune, SAP head, cached trajectories, bs=32, 1 MI355X).

One "episode batch" follows map_nav_src/r2r/agent.py:436-560 without the simulator: language once, then T steps of
  panorama encoder on the current viewpoints -> node-embedding updates of the B topological maps -> per-step map inputs
  (graph_map.GraphMapBatch: batched Floyd graphs, device-resident running means) -> lift + splat of the current viewpoint
  and its visited neighbours straight out of the resident grid-feature store -> navigation mode (global-map encoder,
  BEV encoder, SAP heads, logit fusion).
The observation streams are synthetic walks over random viewpoint graphs (synthetic.make_nav_episodes): "cached
trajectories", i.e. the next viewpoint does not depend on the predicted action.
  --mode infer : torch.no_grad(), eval (validation / test rollouts)
  --mode train : teacher-forced imitation loss summed over the steps, one backward, clip, AdamW (agent.py:339-420,562-600)
Prints one JSON line (ms per navigation step, episodes per second, share of the host-side map bookkeeping).
Not the round's bench.py metric."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vln_bevbert_amd import ops, synthetic  # noqa: E402
from vln_bevbert_amd.config import BevBertConfig  # noqa: E402
from vln_bevbert_amd.feature_store import GridFeatureStore  # noqa: E402
from vln_bevbert_amd.graph_map import GraphMapBatch  # noqa: E402
from vln_bevbert_amd.graph_map_dev import DeviceGraphMap  # noqa: E402
from vln_bevbert_amd.nav_model import VLNBert  # noqa: E402
from vln_bevbert_amd.nav_static import NavGraphRunner, NavTrainRunner  # noqa: E402
from vln_bevbert_amd.pretrain_cmt import bevpos_polar  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=6, help="navigation steps per episode")
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="infer", choices=["infer", "train"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--nodes", type=int, default=14, help="viewpoints per synthetic scan")
    ap.add_argument("--no-graphs", action="store_true",
                    help="infer mode: issue the panorama / navigation forwards eagerly instead of replaying the captured "
                         "steps of nav_static.NavGraphRunner")
    ap.add_argument("--map", default="device", choices=["device", "host"],
                    help="device: graph_map_dev.DeviceGraphMap (maps, relaxation, hop counts, per-step tensors in HIP "
                         "kernels); host: graph_map.GraphMapBatch (batched numpy, rounds 1-3)")
    ap.add_argument("--feedback", action="store_true",
                    help="read the fused logits back every step and pick the action on the host, as the agent does "
                         "(map_nav_src/r2r/agent.py:520-560: nav_probs -> argmax -> .cpu().numpy()): the host's bookkeeping "
                         "of step t+1 can then no longer hide behind the GPU work of step t")
    ap.add_argument("--check", action="store_true",
                    help="size-independent properties of the rollout instead of timing (tests/test_gpu_model.py): every step's "
                         "fused logits are finite or -inf, every live sample has a finite best action, a second rollout "
                         "with the same step seed reproduces the logits")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    cfg = BevBertConfig()
    torch.manual_seed(0)
    model = VLNBert(cfg)
    arena = model.vln_bert.finalize(dev, cdt)
    model.train(a.mode == "train")
    B, T, K = a.batch, a.steps, cfg.bev_dim * cfg.bev_dim
    rng = np.random.default_rng(0)
    g = torch.Generator().manual_seed(0)

    # the "dataset": B scans of `nodes` viewpoints, grid features resident in HBM (fp16)
    keys = [f"scan{i}_e{i}_v{n}" for i in range(B) for n in range(a.nodes)]
    N = len(keys)
    store = GridFeatureStore(keys, torch.randn(N, 12, 196, cfg.grid_feat_size, generator=g).half(),
                             torch.rand(N, 12, 14, 14, generator=g) * 0.6,
                             torch.randint(0, 40, (N, 12, 14, 14), generator=g).to(torch.uint8), dev)
    obs_all, ended_all = synthetic.make_nav_episodes(B, T, seed=1, n_nodes=a.nodes)
    # instructions and per-step panorama features (36 views: candidates first, agent.py:70-112)
    txt_ids = torch.randint(1000, 2000, (B, 80), generator=g).to(dev)
    txt_masks = torch.ones(B, 80, dtype=torch.bool, device=dev)
    pano = []
    for t in range(T):
        nc = [len(ob["candidate"]) for ob in obs_all[t]]
        nav_types = torch.zeros(B, 36, dtype=torch.long)
        for i, n in enumerate(nc):
            nav_types[i, :n] = 1
        ang = torch.rand(B, 36, 2, generator=g) * 6.28
        loc = torch.cat([ang[..., :1].sin(), ang[..., :1].cos(), ang[..., 1:].sin(), ang[..., 1:].cos(),
                         torch.ones(B, 36, 3)], -1)
        pano.append({"view_img_fts": torch.randn(B, 36, cfg.image_feat_size, generator=g).to(dev), "obj_img_fts": None,
                     "loc_fts": loc.to(dev), "nav_types": nav_types.to(dev),
                     "view_lens": torch.full((B,), 36, dtype=torch.long, device=dev), "obj_lens": None})
    pix, polar = ops.pixel_scale(cfg.grid_hw, dev), bevpos_polar(cfg.bev_dim, dev)
    t_book = [0.0]
    # infer: captured forward steps (nav_static.NavGraphRunner); train: forward + backward graphs per step of the episode
    # (nav_static.NavTrainRunner; --no-graphs runs the same segments eagerly)
    runner = NavGraphRunner(model) if a.mode == "infer" else NavTrainRunner(model, arena, graphs=not a.no_graphs)
    use_graphs = [a.mode == "train" or not a.no_graphs]
    if a.mode == "train":
        a.warmup = max(a.warmup, runner.eager_uses + 1)      # eager episodes + the capture episode stay out of the timed region

    trace = []

    on_device = a.map == "device"
    actions = []

    runner_box = [runner]

    def episode():
        start = [ob["viewpoint"] for ob in obs_all[0]]
        gm = DeviceGraphMap(start, cfg.hidden_size, dev, dtype=cdt) if on_device else \
            GraphMapBatch(start, cfg.hidden_size, dev, dtype=cdt)
        if not on_device:
            gm.update_graph(obs_all[0])
        txt = model("language", {"txt_ids": txt_ids, "txt_masks": txt_masks})
        loss = 0.0
        for t in range(T):
            obs, ended = obs_all[t], ended_all[t]
            h0 = time.perf_counter()
            keys = [f"{ob['scan']}_{ob['viewpoint']}" for ob in obs]
            if on_device:       # graph update + step ids + store rows / poses of the new viewpoints: one launch
                gm.update_graph(obs, None if t == 0 else ended_all[t - 1], step_id=t + 1, step_ended=ended,
                                store_rows=[store.row[k] for k in keys])
            else:
                if t > 0:
                    gm.update_graph(obs, ended_all[t - 1])
                gm.set_step_ids(obs, t, ended)
            t_book[0] += time.perf_counter() - h0
            runner = runner_box[0]
            pe, pm = runner.panorama(pano[t]) if use_graphs[0] else model("panorama", pano[t])
            avg = (pe * pm[..., None]).sum(1) / pm.sum(1, keepdim=True)                       # agent.py:478-479
            h0 = time.perf_counter()
            gm.update_node_embeds(obs, None if on_device else [[c["viewpointId"] for c in ob["candidate"]] for ob in obs],
                                  avg, pe, ended)          # the device map reads the candidate ids from the observations
            if not on_device:
                gm.remember_views(obs, keys, store, ended)
            nav = gm.nav_gmap_variable(obs)
            bi = gm.bev_inputs(obs, store, pc_order=1, bev_dim=cfg.bev_dim, bev_res=cfg.bev_res)
            t_book[0] += time.perf_counter() - h0
            _, order, start = ops.bev_lift_bin(bi["depths"], bi["T_c2w"], bi["T_w2c"], bi["S_w2c"], pix, cfg.bev_dim,
                                               cfg.bev_res)
            bev_fts, _, _ = ops.bev_splat_mean(store.rgbs, order, start, K, out_dtype=cdt, rows=bi["grid_rows"])
            nav.update({
                "txt_embeds": txt, "txt_masks": txt_masks, "bev_fts": bev_fts,
                "bev_pos_fts": torch.cat([bi["bev_gpos_fts"].expand(-1, K, -1), polar[None].expand(B, -1, -1)], -1),
                "bev_masks": torch.ones(B, K, dtype=torch.bool, device=dev), "bev_nav_masks": bi["bev_nav_masks"],
                "bev_cand_idxs": bi["bev_cand_idxs"], "bev_cand_vpids": bi["bev_cand_vpids"], "obj_embeds": None,
                "obj_masks": None})
            out = runner.navigation(nav) if use_graphs[0] else model("navigation", nav)
            if a.feedback:      # agent.py:520-560: the action is picked on the host from the logits of THIS step
                probs = torch.softmax(out["fused_logits"].float(), 1)
                a_t = probs.argmax(1).cpu().numpy()                     # device -> host: the step's synchronisation point
                actions.append([nav["gmap_vpids"][i][int(k)] for i, k in enumerate(a_t)])
            if a.check:
                trace.append(out["fused_logits"].float().clone())
            if a.mode == "train":       # teacher action: [stop] is always a valid target of the fused logits
                live = torch.from_numpy(~ended).to(dev)
                tgt = torch.zeros(B, dtype=torch.long, device=dev)
                loss = loss + (F.cross_entropy(out["fused_logits"].float(), tgt, reduction="none") * live).sum()
        return loss

    last_loss = [None]
    grads_only = [False]

    def iteration(i):
        ops.RT.new_step(1000 + i)
        if a.mode == "infer":
            with torch.no_grad():
                episode()
        else:
            rn = runner_box[0]
            rn.begin_episode()
            if rn.capturing:                # the capture episode: forward graphs in rollout order, then the backward graphs
                episode()                   # in reverse; nothing of its backward executes, so it does not train
                rn.end_capture()
                if rn.graph_error is not None:
                    print("capture failed:", rn.graph_error, getattr(rn, "graph_traceback", ""), file=sys.stderr, flush=True)
                return
            arena.zero_grad()
            loss = episode() / B
            loss.backward()
            arena.sync()
            if grads_only[0]:
                last_loss[0] = loss.detach()
                return
            arena.clip_and_step(1e-5, max_norm=40.0)
            last_loss[0] = loss.detach()

    if a.check and a.mode == "train":
        # captured training rollout against the same segments issued eagerly: from the same parameters and optimiser state,
        # two training episodes, then the gradients of a third -- the arena must agree BIT FOR BIT (same kernels, same
        # arguments, the deferred weight-gradient / reduction work flushed at the same points in the same order)
        iteration(0)                                         # primes the optimiser state (and the library's GEMM plans)
        torch.cuda.synchronize()
        keep = {k: getattr(arena, k).clone() for k in ("params", "exp_avg", "exp_avg_sq", "chunk_steps")}
        got = {}
        for name, graphs in (("eager", False), ("graphs", True)):
            for k, v in keep.items():
                getattr(arena, k).copy_(v)
            arena.sync_shadow()
            runner_box[0] = NavTrainRunner(model, arena, graphs=graphs)
            seeds = [11, 12] + ([13] if graphs else []) + [14]          # (the capture episode of the graph arm does not train)
            for j, sd in enumerate(seeds):
                grads_only[0] = j == len(seeds) - 1
                iteration(sd)
            torch.cuda.synchronize()
            got[name] = (arena.grads.clone(), float(last_loss[0]), dict(runner_box[0].stats), runner_box[0].captured_graphs(),
                         runner_box[0].graph_error)
            grads_only[0] = False
        g0, g1 = got["eager"][0], got["graphs"][0]
        same = bool(torch.equal(g0, g1))
        rel = float((g0.double() - g1.double()).norm() / g0.double().norm().clamp_min(1e-30))
        print(json.dumps({"check": "ok" if (same and got["graphs"][3] > 0 and got["graphs"][4] is None) else "FAILED",
                          "mode": "train", "batch": B, "steps": T, "dtype": a.dtype, "gradients_bitwise_equal": same,
                          "gradient_rel_l2_diff": rel, "loss_eager": got["eager"][1], "loss_graphs": got["graphs"][1],
                          "captured_graphs": got["graphs"][3], "runner": got["graphs"][2], "graph_error": got["graphs"][4],
                          "grad_norm": float(g0.double().norm())}))
        return
    if a.check:
        runs = []
        for _ in range(2):
            trace.clear()
            iteration(7)
            torch.cuda.synchronize()
            runs.append([t.cpu() for t in trace])
        assert len(runs[0]) == T
        for t in range(T):
            x, y = runs[0][t], runs[1][t]
            assert not torch.isnan(x).any() and not (x == float("inf")).any(), t
            live = torch.from_numpy(~ended_all[t])
            assert bool(torch.isfinite(x.max(1).values[live]).all()), t          # a live sample can always act ([stop])
            fin = torch.isfinite(x)
            assert torch.equal(fin, torch.isfinite(y)) and float((x[fin] - y[fin]).abs().max()) <= 1e-6 * max(1.0, float(x[fin].abs().max())), t
        rec = {"check": "ok", "batch": B, "steps": T, "dtype": a.dtype, "map_nodes_last_step": int(runs[0][-1].shape[1])}
        if use_graphs[0]:
            # the captured steps against the eager forwards on the same inputs: node / candidate padding and the static
            # buffers must not change which actions are possible, nor the logits beyond the rounding of differently
            # shaped GEMMs.  Four rollouts, so that every bucket has passed its eager uses and is replayed.
            for k in range(3):
                trace.clear()
                iteration(7)
            torch.cuda.synchronize()
            graphed = [t.cpu() for t in trace]
            assert runner.graph_error is None, runner.graph_error
            assert runner.captured_graphs() >= 2 and runner.stats["replays"] > 0, runner.stats
            use_graphs[0] = False
            trace.clear()
            iteration(7)
            torch.cuda.synchronize()
            eager = [t.cpu() for t in trace]
            tol = 3e-2 if a.dtype == "bf16" else 1e-4
            worst = 0.0
            for t in range(T):
                x, y = graphed[t], eager[t]
                assert x.shape == y.shape, (t, x.shape, y.shape)
                fin = torch.isfinite(y)
                assert torch.equal(torch.isfinite(x), fin), t
                worst = max(worst, float((x[fin] - y[fin]).abs().max()) / max(1.0, float(y[fin].abs().max())))
                live = torch.from_numpy(~ended_all[t])
                same = (x.argmax(1) == y.argmax(1)) | ~live
                # an argmax may flip only between logits that are closer than the tolerance
                flip = (~same).nonzero().flatten().tolist()
                for i in flip:
                    assert abs(float(y[i].max() - y[i, x[i].argmax()])) <= tol * max(1.0, float(y[i][fin[i]].abs().max())), (t, i)
            assert worst <= tol, worst
            rec.update({"graphs_vs_eager_max_rel_diff": round(worst, 5), "captured_graphs": runner.captured_graphs(),
                        "runner": runner.stats})
        print(json.dumps(rec))
        return
    for i in range(a.warmup):
        iteration(i)
    torch.cuda.synchronize()
    t_book[0] = 0.0
    t0 = time.perf_counter()
    for i in range(a.iters):
        iteration(a.warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    overflow = None
    if on_device:
        chk = DeviceGraphMap([ob["viewpoint"] for ob in obs_all[0]], cfg.hidden_size, dev, dtype=cdt)
        for t in range(T):      # the neighbour bound of bev_inputs held on every step of these episodes
            obs = obs_all[t]
            chk.update_graph(obs, None if t == 0 else ended_all[t - 1], step_id=t + 1, step_ended=ended_all[t],
                             store_rows=[store.row[f"{ob['scan']}_{ob['viewpoint']}"] for ob in obs])
            chk.nav_gmap_variable(obs)
            chk.bev_inputs(obs, store, pc_order=1, bev_dim=cfg.bev_dim, bev_res=cfg.bev_res)
        overflow = chk.check_overflow()
    print(json.dumps({"workload": f"R2R fine-tune rollout, {a.mode}, batch {B}, {T} navigation steps, {a.dtype}, "
                                  f"{'DeviceGraphMap (maps on the device)' if on_device else 'GraphMapBatch (host numpy)'} + "
                                  f"resident grid-feature store ({store.nbytes() / 2 ** 30:.1f} GiB)",
                      "action_feedback": "logits read back and argmaxed on the host every step (agent.py:520-560)"
                      if a.feedback else "none (cached trajectories, no device->host copy inside the episode)",
                      "map": a.map, "neighbour_bound_overflow": overflow,
                      "ms_per_episode_batch": round(dt * 1e3, 2), "ms_per_nav_step": round(dt * 1e3 / T, 2),
                      "episodes_per_s": round(B / dt, 1), "nav_steps_per_s": round(B * T / dt, 1),
                      "host_map_bookkeeping_ms_per_nav_step": round(t_book[0] / a.iters / T * 1e3, 2),
                      "final_loss": None if last_loss[0] is None else round(float(last_loss[0]), 5),
                      "text_kv_cache": bool(use_graphs[0] and runner.text_cache and any("g_kv" in v for v in runner.shared.values())),
                      "step_launch": ("hipGraph replay per mode and shape bucket (%d graphs, %d replays, %d eager calls)"
                                      % (runner.captured_graphs(), runner.stats["replays"], runner.stats["eager"]))
                      if (use_graphs[0] and runner.captured_graphs()) else "eager",
                      "graph_error": runner.graph_error if runner is not None else None}))


if __name__ == "__main__":
    main()
