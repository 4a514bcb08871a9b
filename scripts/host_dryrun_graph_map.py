#!/usr/bin/env python3
"""Host time of graph_map_dev.DeviceGraphMap per navigation step with the device stubbed out (CPU only): the C-ABI calls
return at once, tensors are host tensors, HostFeed takes its CPU path.  What remains is the Python the rollout loop pays
per step: id -> node-index lookups, the node ordering, the id lists, packing the step record.
    python scripts/host_dryrun_graph_map.py [batch=32] [steps=15]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vln_bevbert_amd import graph_map_dev, lib, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 15
H, NODES = 768, 14
lib.call = lambda *a: None
lib.stream = lambda: 0
_empty = torch.empty
torch.empty = lambda *a, **k: torch.zeros(*a, **k)      # outputs the stubbed kernels would have written


class _Dry(graph_map_dev.DeviceGraphMap):
    def __init__(self, start_vps, hidden_size):          # DeviceGraphMap's host state on a CPU device (it refuses one itself)
        self._setup(start_vps, hidden_size, torch.device("cpu"), torch.float32, 64, 12)


class _Store:
    V, hw = 12, 14
    row = {f"scan{i}_e{i}_v{n}": i * NODES + n for i in range(B) for n in range(NODES)}
    depths = torch.zeros(B * NODES, 12, 14, 14)


obs_all, ended_all = synthetic.make_nav_episodes(B, T, seed=1, n_nodes=NODES)
g = torch.Generator().manual_seed(0)
avg = [torch.randn(B, H, generator=g) for _ in range(T)]
pano = [torch.randn(B, 36, H, generator=g) for _ in range(T)]
tm = {}


def tick(name, t0):
    tm[name] = tm.get(name, 0.0) + time.perf_counter() - t0


def episode():
    gm = _Dry([ob["viewpoint"] for ob in obs_all[0]], H)
    for t in range(T):
        obs, ended = obs_all[t], ended_all[t]
        t0 = time.perf_counter()
        keys = [f"{ob['scan']}_{ob['viewpoint']}" for ob in obs]
        gm.update_graph(obs, None if t == 0 else ended_all[t - 1], step_id=t + 1, step_ended=ended,
                        store_rows=[_Store.row[k] for k in keys])
        tick("update_graph(+step ids, store rows)", t0); t0 = time.perf_counter()
        gm.update_node_embeds(obs, None, avg[t], pano[t], ended)
        tick("update_node_embeds", t0); t0 = time.perf_counter()
        gm.nav_gmap_variable(obs)
        tick("nav_gmap_variable", t0); t0 = time.perf_counter()
        gm.bev_inputs(obs, _Store, pc_order=1)
        tick("bev_inputs", t0)


episode()
runs = []
for _ in range(20):
    tm.clear()
    episode()
    runs.append(dict(tm))
best = min(runs, key=lambda r: sum(r.values()))        # the quietest of 20 episodes: this host is shared
for k, v in best.items():
    print(f"{k:40s} {v / T * 1e3:7.3f} ms per navigation step")
print(f"{'total':40s} {sum(best.values()) / T * 1e3:7.3f} ms per navigation step (batch {B}, torch CPU tensors, this host; "
      f"median of 20 episodes {np.median([sum(r.values()) for r in runs]) / T * 1e3:.3f})")
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    episode()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
