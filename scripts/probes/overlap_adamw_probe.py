#!/usr/bin/env python3
"""Would a deferred AdamW of the map-encoder half of the parameters hide behind the NEXT step's text-encoder forward?
The text encoder's 5 120-row kernels are latency-bound and leave the HBM idle; AdamW is pure bandwidth.  Measures the
replayed text-encoder forward (batch 64, bf16) alone, and with an AdamW pass over 136 M parameters running beside it on
another stream (default and lowest stream priority)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vln_bevbert_amd import ops, synthetic  # noqa: E402
from vln_bevbert_amd.config import BevBertConfig  # noqa: E402
from vln_bevbert_amd.lib import call, ptr  # noqa: E402
from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining  # noqa: E402

cfg = BevBertConfig()
torch.manual_seed(0)
model = GlocalTextPathCMTPreTraining(cfg)
arena = model.finalize("cuda", torch.bfloat16)
model.eval()
b = synthetic.batch_to(synthetic.make_batch(cfg, "mlm", 64, seed=1, sems_as="ids"), "cuda")
with torch.no_grad():
    for _ in range(3):
        model.bert._text(b["txt_ids"], b["txt_lens"])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        model.bert._text(b["txt_ids"], b["txt_lens"])
first_map = min(arena.slices[n][0] for n in arena.slices if n.startswith(("bert.local_encoder", "bert.global_encoder")) or not n.startswith("bert."))
n2 = arena.numel - first_map
a = arena
a.exp_avg, a.exp_avg_sq = torch.zeros_like(a.params), torch.zeros_like(a.params)
a._flags_host[:] |= 2
a._flags_dirty = True
a.upload_flags()
a.set_lr(1e-5)
a._scalars[1] = 1.0


def adamw_region2(stream):
    o, c = first_map, first_map // 1024
    call("bevbert_adamw_step", a.params[o:].data_ptr(), a.grads[o:].data_ptr(), a.exp_avg[o:].data_ptr(),
         a.exp_avg_sq[o:].data_ptr(), a.shadow[o:].data_ptr(), a.flags[c:].data_ptr(), a.chunk_steps[c:].data_ptr(), n2,
         a._scalars[1:].data_ptr(), a._scalars[2:].data_ptr(), 0.0, 0.9, 0.98, 1e-6, 0.01, stream.cuda_stream)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


main = torch.cuda.current_stream()
print(f"text encoder forward alone: {timed(g.replay):.3f} ms")
print(f"AdamW over {n2 / 1e6:.0f} M parameters alone: {timed(lambda: adamw_region2(main)):.3f} ms")
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
for label, side in (("default-priority side stream", torch.cuda.Stream()), ("lowest-priority side stream", torch.cuda.Stream(priority=0))):
    def both():
        side.wait_stream(main)
        adamw_region2(side)
        g.replay()
        main.wait_stream(side)
    print(f"text forward + AdamW beside it ({label}): {timed(both):.3f} ms   (sum of the two alone above)")
hp = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(hp):
    g2 = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g2, capture_error_mode="thread_local"):
        model.bert._text(b["txt_ids"], b["txt_lens"])
side = torch.cuda.Stream(priority=0)


def both_hp():
    with torch.cuda.stream(hp):
        side.wait_stream(hp)
        adamw_region2(side)
        g2.replay()
        hp.wait_stream(side)


torch.cuda.synchronize()
with torch.cuda.stream(hp):
    print(f"the same with the text forward on a HIGH-priority stream: {timed(both_hp):.3f} ms")
