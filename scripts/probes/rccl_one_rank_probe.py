#!/usr/bin/env python3
"""Which collectives of the bf16 gradient exchange survive on a ONE-rank RCCL group (the only RCCL configuration a
single-GPU box can run)?  Prints one line per collective before calling it, so a crash names its cause."""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.randn(1 << 20, device="cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "allreduce"):
    print("all_reduce fp32", flush=True)
    dist.all_reduce(x)
    torch.cuda.synchronize()
s = x.to(torch.bfloat16)
r = torch.empty_like(s)
if which in ("all", "a2a"):
    print("all_to_all_single bf16", flush=True)
    dist.all_to_all_single(r, s)
    torch.cuda.synchronize()
if which in ("all", "ag"):
    print("all_gather_into_tensor bf16", flush=True)
    dist.all_gather_into_tensor(r, s)
    torch.cuda.synchronize()
print("ok", flush=True)
dist.destroy_process_group()
