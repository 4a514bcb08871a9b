"""graph_map.HostFeed against per-array torch.from_numpy(a).to(device) copies: host time per call with an idle GPU and
behind a queued 7 ms matmul per iteration.  In the second case BOTH block for the matmul (the loop is GPU-bound and
the ring is 8 slots deep) -- what the packed feed buys shows in the idle case (one copy instead of five) and in the
rollout bench, where the GPU is not saturated.  usage: python scripts/probes/hostfeed_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vln_bevbert_amd.graph_map import HostFeed
dev = torch.device("cuda", 0)
f = HostFeed(dev)
arrs = {"a": np.zeros((32, 16), np.int64), "b": np.zeros((32, 16, 7), np.float32), "c": np.zeros((32, 16), bool),
        "d": np.zeros((32, 16, 16), np.float32), "e": np.zeros((32, 48, 4, 4), np.float32)}
for name, busy in (("idle", False), ("busy", True)):
    x = torch.randn(8192, 8192, device=dev)
    torch.cuda.synchronize()
    ts = []
    for i in range(40):
        if busy:
            y = x @ x
        t0 = time.perf_counter()
        out = f(arrs)
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print(name, "HostFeed call us: median %.1f max %.1f" % (np.median(ts) * 1e6, max(ts) * 1e6))
    ts = []
    for i in range(40):
        if busy:
            y = x @ x
        t0 = time.perf_counter()
        out = {k: torch.from_numpy(v).to(dev) for k, v in arrs.items()}
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print(name, "per-array .to(dev) us: median %.1f max %.1f" % (np.median(ts) * 1e6, max(ts) * 1e6))
