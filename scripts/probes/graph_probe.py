#!/usr/bin/env python3
"""Feasibility probe for capturing the training step in a hipGraph (run on the GPU box):
  1. torch.nonzero_static on the device, eagerly and under capture;
  2. fwd + bwd of a text encoder layer stack (C-ABI kernels, direct hipBLASLt, the weight-gradient side stream with its
     pooled events) captured with torch.cuda.graph and replayed: gradients must equal the eager run bit for bit;
  3. a one-rank RCCL all_reduce on a side stream inside the capture."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vln_bevbert_amd import ops, weights  # noqa: E402
from vln_bevbert_amd.config import BevBertConfig  # noqa: E402
from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining  # noqa: E402

dev = torch.device("cuda", 0)


def section(name):
    print(f"\n=== {name}", flush=True)


section("1. nonzero_static")
try:
    m = torch.rand(1000, device=dev) < 0.2
    idx = torch.nonzero_static(m, size=400, fill_value=0)
    print("eager ok", idx.shape, int(m.sum()), int((idx[: int(m.sum())].squeeze(1) == m.nonzero().squeeze(1)).all()))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out = torch.nonzero_static(m, size=400, fill_value=0)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = torch.nonzero_static(m, size=400, fill_value=0)
    m.copy_(torch.rand(1000, device=dev) < 0.3)
    g.replay()
    torch.cuda.synchronize()
    n = int(m.sum())
    print("capture ok", bool((out[:min(n, 400)].squeeze(1) == m.nonzero().squeeze(1)[:400]).all()))
except Exception as e:
    print("nonzero_static FAILED:", repr(e))

section("2. fwd+bwd of the text encoder under capture (wgrad side stream on)")
cfg = BevBertConfig.tiny(num_l_layers=2, num_x_layers=1, vocab_size=400)
model = GlocalTextPathCMTPreTraining(cfg)
model.load_state_dict(weights.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model.tie_weights()
arena = model.finalize(dev, torch.bfloat16)
model.train()
model.set_dropout(0.1)
ids = torch.randint(5, 300, (8, 40), device=dev)
lens = torch.full((8,), 40, device=dev)


def step():
    ops.RT.new_step(123)
    arena.zero_grad()
    out, _ = model.bert._text(ids, lens)
    loss = out.float().pow(2).mean()
    loss.backward()
    arena.sync()
    return loss


try:
    for _ in range(3):
        l_eager = step()
    torch.cuda.synchronize()
    g_eager = arena.grads.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    with torch.cuda.graph(graph):
        l_cap = step()
    torch.cuda.synchronize()
    print(f"captured in {time.perf_counter() - t0:.2f} s")
    arena.grads.fill_(7.0)
    graph.replay()
    torch.cuda.synchronize()
    print("replay loss", float(l_cap), "eager loss", float(l_eager))
    d = (arena.grads - g_eager).abs().max()
    print("max |grad_replay - grad_eager| =", float(d), " grads nonzero:", float(g_eager.abs().sum()) > 0)
    t0 = time.perf_counter()
    for _ in range(20):
        graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"20 replays: host {1e3 * (t1 - t0) / 20:.3f} ms each, wall {1e3 * (t2 - t0) / 20:.3f} ms each")
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"20 eager  : host {1e3 * (t1 - t0) / 20:.3f} ms each, wall {1e3 * (t2 - t0) / 20:.3f} ms each")
except Exception as e:
    import traceback
    traceback.print_exc()
    print("capture FAILED:", repr(e))
    torch.cuda.synchronize()

section("3. one-rank RCCL all_reduce inside a capture")
try:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    buf = torch.ones(1 << 20, device=dev)
    dist.all_reduce(buf)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        buf.mul_(2.0)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            w = dist.all_reduce(buf, async_op=True)
        w.wait()
        torch.cuda.current_stream().wait_stream(side)
        buf.add_(1.0)
    buf.fill_(1.0)
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    print("rccl-in-capture ok; buf[0] =", float(buf[0]), "(expect 7)")
    dist.destroy_process_group()
except Exception as e:
    import traceback
    traceback.print_exc()
    print("rccl capture FAILED:", repr(e))
print("\nprobe done")
