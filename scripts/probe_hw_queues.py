#!/usr/bin/env python3
"""Is the hardware queue of a HIP stream a fixed property?  Probes the 32 streams of torch's pool against the compute stream
(hwqueues.shares_hw_queue) three times: fresh, after work on all of them, and after a captured graph with side branches has
been instantiated and replayed.  One line per pass: the pool positions that share the compute stream's queue."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vln_bevbert_amd.hwqueues import shares_hw_queue  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
main = torch.cuda.current_stream(dev)
pool = [torch.cuda.Stream(dev) for _ in range(32)]


def scan(tag):
    got = [i for i, st in enumerate(pool) if shares_hw_queue(st, main, dev)]
    print(f"{tag:48s} shares the compute queue: {got}", flush=True)
    return got


a = scan("fresh")
b = scan("again")
x = torch.randn(4096, 4096, device=dev)
for st in pool:
    with torch.cuda.stream(st):
        (x @ x).sum()
torch.cuda.synchronize()
c = scan("after a GEMM on every stream")
g = torch.cuda.CUDAGraph()
s1, s2, s3 = pool[1], pool[2], pool[3]
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    for s in (s1, s2, s3):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            y = x @ x
    z = x @ x
    for s in (s1, s2, s3):
        cur.wait_stream(s)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
d = scan("after a 4-branch graph was captured and replayed")
side = torch.cuda.Stream(dev)          # what the default stream looks like from a non-default compute stream
with torch.cuda.stream(side):
    main2 = torch.cuda.current_stream(dev)
    e = [i for i, st in enumerate(pool) if st.cuda_stream != main2.cuda_stream and shares_hw_queue(st, main2, dev)]
print(f"{'against pool stream 0 (wrapped) as compute stream':48s} shares its queue: {e}")
print("stable" if a == b == c == d else "NOT stable")
