#!/usr/bin/env python3
"""Fine-tune rollout timing (BASELINE.json configs[4]: R2R fine-tune, SAP head, cached trajectories, bs=32, 1 MI355X).

One "episode batch" = language once, then T navigation steps; every step runs the panorama encoder on the current
viewpoints, lifts + splats the 1-hop grid features into the BEV and runs the navigation mode (global map encoder, BEV
encoder, SAP heads, logit fusion) -- map_nav_src/r2r/agent.py:194-337 without the simulator and the GraphMap bookkeeping
(cached trajectories: every step's inputs come from one synthetic pre-training-shaped batch).
  --mode infer : torch.no_grad(), eval (validation / test rollouts)
  --mode train : teacher-forced imitation loss summed over the steps, one backward, clip, AdamW (agent.py:339-420)
Prints one JSON line (ms per navigation step, episodes per second).  Not the round's bench.py metric."""
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
from vln_bevbert_amd.nav_model import VLNBert  # noqa: E402
from vln_bevbert_amd.pretrain_cmt import bevpos_polar  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5, help="navigation steps per episode")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--mode", default="infer", choices=["infer", "train"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    cfg = BevBertConfig()
    torch.manual_seed(0)
    model = VLNBert(cfg)
    arena = model.vln_bert.finalize(dev, cdt)
    model.train(a.mode == "train")
    B, T = a.batch, a.steps
    pb = synthetic.make_batch(cfg, "sap", B, seed=5, n_steps=T, sems_as="ids")
    d = synthetic.batch_to(pb, dev)
    K = cfg.bev_dim * cfg.bev_dim
    pix, polar = ops.pixel_scale(cfg.grid_hw, dev), bevpos_polar(cfg.bev_dim, dev)
    txt_masks = torch.arange(d["txt_ids"].shape[1], device=dev)[None] < d["txt_lens"][:, None]
    starts = np.cumsum([0] + pb["traj_step_lens"][:-1])
    G = int(pb["gmap_lens"].max())
    gmasks = torch.arange(G, device=dev)[None] < d["gmap_lens"][:, None]
    cand_vpids = [[None] + c[-1] for c in pb["traj_cand_vpids"]]
    feat = d["rgbs"].reshape(B, -1, d["rgbs"].shape[-1])
    bev_pos = torch.cat([d["bev_gpos_fts"].expand(-1, K, -1), polar[None].expand(B, -1, -1)], -1)

    def episode():
        txt = model("language", {"txt_ids": d["txt_ids"], "txt_masks": txt_masks})
        loss = 0.0
        for t in range(T):
            rows = torch.from_numpy(starts + t).to(dev)
            pano, pmask = model("panorama", {"view_img_fts": d["traj_view_img_fts"][rows], "obj_img_fts": None,
                                             "loc_fts": d["traj_loc_fts"][rows], "nav_types": d["traj_nav_types"][rows],
                                             "view_lens": d["traj_vp_view_lens"][rows], "obj_lens": None})
            # GraphMap.update_node_embed / get_node_embed: the visited node's embedding is the mean of its views
            node = (pano * pmask[..., None]).sum(1) / pmask.sum(1, keepdim=True)
            gimg = node[:, None].expand(-1, G, -1).contiguous()
            _, order, start = ops.bev_lift_bin(d["depths"], d["T_c2w"], d["T_w2c"], d["S_w2c"], pix, cfg.bev_dim,
                                               cfg.bev_res)
            bev_fts, _, _ = ops.bev_splat_mean(feat, order, start, K, out_dtype=cdt)
            out = model("navigation", {
                "txt_embeds": txt, "txt_masks": txt_masks, "gmap_img_embeds": gimg,
                "gmap_step_ids": d["gmap_step_ids"], "gmap_pos_fts": d["gmap_pos_fts"], "gmap_masks": gmasks,
                "gmap_pair_dists": d["gmap_pair_dists"], "gmap_visited_masks": d["gmap_visited_masks"],
                "gmap_visited_masks_cpu": d.get("gmap_visited_masks_cpu"),
                "gmap_vpids": pb["gmap_vpids"], "bev_fts": bev_fts, "bev_pos_fts": bev_pos,
                "bev_masks": torch.ones(B, K, dtype=torch.bool, device=dev), "bev_nav_masks": d["bev_nav_masks"],
                "bev_cand_idxs": d["bev_cand_idxs"], "bev_cand_vpids": cand_vpids, "obj_embeds": None,
                "obj_masks": None})
            if a.mode == "train":
                loss = loss + F.cross_entropy(out["fused_logits"].float(), d["global_act_labels"], reduction="sum")
        return loss

    def iteration(i):
        ops.RT.new_step(1000 + i)
        if a.mode == "infer":
            with torch.no_grad():
                episode()
        else:
            arena.zero_grad()
            (episode() / B).backward()
            arena.clip_and_step(1e-5, max_norm=40.0)

    for i in range(a.warmup):
        iteration(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.iters):
        iteration(a.warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print(json.dumps({"workload": f"R2R fine-tune rollout, {a.mode}, batch {B}, {T} navigation steps, {a.dtype}",
                      "ms_per_episode_batch": round(dt * 1e3, 2), "ms_per_nav_step": round(dt * 1e3 / T, 2),
                      "episodes_per_s": round(B / dt, 1), "nav_steps_per_s": round(B * T / dt, 1)}))


if __name__ == "__main__":
    main()
