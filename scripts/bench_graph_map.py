#!/usr/bin/env python3
"""Host cost of the fine-tune map bookkeeping per navigation step (CPU only): graph_map.GraphMapBatch against the
reference's GraphMap / FloydGraph driven the same way (the latter only where /root/reference is mounted: the build
container).  Same synthetic observation streams, batch 32."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vln_bevbert_amd import synthetic  # noqa: E402
from vln_bevbert_amd.graph_map import GraphMapBatch  # noqa: E402

B, T, H, NODES = 32, 7, 768, 14
obs_all, ended_all = synthetic.make_nav_episodes(B, T, seed=1, n_nodes=NODES)
g = torch.Generator().manual_seed(0)
avg = [torch.randn(B, H, generator=g) for _ in range(T)]
pano = [torch.randn(B, 36, H, generator=g) for _ in range(T)]


class _Store:
    V, hw = 12, 14
    row = {f"scan{i}_e{i}_v{n}": i * NODES + n for i in range(B) for n in range(NODES)}
    depths = torch.zeros(B * NODES, 12, 14, 14)


def ours():
    gm = GraphMapBatch([ob["viewpoint"] for ob in obs_all[0]], H, "cpu")
    gm.update_graph(obs_all[0])
    for t in range(T):
        obs, ended = obs_all[t], ended_all[t]
        if t > 0:
            gm.update_graph(obs, ended_all[t - 1])
        gm.set_step_ids(obs, t, ended)
        gm.update_node_embeds(obs, [[c["viewpointId"] for c in ob["candidate"]] for ob in obs], avg[t], pano[t], ended)
        gm.remember_views(obs, [f"{ob['scan']}_{ob['viewpoint']}" for ob in obs], _Store, ended)
        gm.nav_gmap_variable(obs)
        gm.bev_inputs(obs, _Store, pc_order=1)


def reference():
    from models import graph_utils
    from oracle import graph_ref
    gmaps = [graph_utils.GraphMap(ob["viewpoint"]) for ob in obs_all[0]]
    for i, ob in enumerate(obs_all[0]):
        gmaps[i].update_graph(ob)
    for t in range(T):
        obs, ended = obs_all[t], ended_all[t]
        if t > 0:
            for i, ob in enumerate(obs):
                if not ended_all[t - 1][i]:
                    gmaps[i].update_graph(ob)
        for i, gm in enumerate(gmaps):
            if not ended[i]:
                gm.node_step_ids[obs[i]["viewpoint"]] = t + 1
                vp = obs[i]["viewpoint"]
                gm.update_node_embed(vp, avg[t][i], rewrite=True)
                gm.update_node_pc(vp, torch.zeros(1, 3), torch.zeros(1, dtype=torch.bool), torch.zeros(1, 1))
                for j, cc in enumerate(obs[i]["candidate"]):
                    if not gm.graph.visited(cc["viewpointId"]):
                        gm.update_node_embed(cc["viewpointId"], pano[t][i, j])
        graph_ref.nav_gmap_variable(obs, gmaps)
        for ob, gm in zip(obs, gmaps):
            gm.gather_node_pc(ob["viewpoint"], 1)
            gm.get_pos_fts(ob["viewpoint"], [gm.start_vp], ob["heading"], ob["elevation"])


def bench(fn, n=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n / T * 1e3


print(f"graph_map.GraphMapBatch   : {bench(ours):7.2f} ms per navigation step (batch {B}, host, torch CPU tensors)")
if os.path.isdir("/root/reference/map_nav_src"):
    import types
    sys.modules.setdefault("torch_scatter", types.ModuleType("torch_scatter"))
    sys.path.insert(0, "/root/reference/map_nav_src")
    print(f"reference GraphMap (x{B})   : {bench(reference):7.2f} ms per navigation step (without its point-cloud "
          f"concatenation and pad_tensors collation)")
