#!/usr/bin/env python3
"""Which parameter gradients differ between two identical backward passes (same seed, same step)?  Diagnostic for
tests/test_gpu_model.py::test_full_size_batch_properties (round 4: r2r_b64 mlm showed ~40 differing elements)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vln_bevbert_amd import ops, synthetic  # noqa: E402
from vln_bevbert_amd.config import BevBertConfig  # noqa: E402
from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "mlm"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pre = sys.argv[3].split(",") if len(sys.argv) > 3 else []          # tasks to run first in the same process (as the test does)
cfg, B, L = BevBertConfig(), 64, 80
torch.manual_seed(0)
model = GlocalTextPathCMTPreTraining(cfg)
arena = model.finalize("cuda", torch.bfloat16)
model.train()
model.set_dropout(0.1)
import warnings
warnings.simplefilter("always")
for t in pre:
    bb = synthetic.batch_to(synthetic.make_batch(cfg, t, B, seed=2000, txt_len=L, sems_as="ids"), "cuda")
    for _ in range(3):
        ops.RT.new_step(77)
        arena.zero_grad()
        model(bb, t).mean().backward()
        arena.sync()
    torch.cuda.synchronize()
b = synthetic.batch_to(synthetic.make_batch(cfg, task, B, seed=2000, txt_len=L, sems_as="ids"), "cuda")
# the gradient row handed to the 2-row token-type table by the panorama branch (one row: the broadcast sum torch computed)
seen_d = []
_orig_small = ops.embedding_grad_small


def _spy(ids, d, sink, table_rows):
    if table_rows == 2 and d.shape[0] == 1:
        seen_d.append(d.detach().float().clone())
    return _orig_small(ids, d, sink, table_rows)


ops.embedding_grad_small = _spy
import vln_bevbert_amd.vilmodel as _vm  # noqa: E402
runs = []
for rep in range(reps + 1):
    ops.RT.new_step(77)
    arena.zero_grad()
    loss = model(b, task)
    loss.mean().backward()
    arena.sync()
    torch.cuda.synchronize()
    runs.append(arena.grads.clone())
base = runs[1]
for rep in range(2, reps + 1):
    d = runs[rep] != base
    n = int(d.sum())
    print(f"rep {rep} vs 1: {n} differing elements (WGRAD_STREAMS={os.environ.get('BEVBERT_WGRAD_STREAMS')}, "
          f"WGRAD_STREAM={os.environ.get('BEVBERT_WGRAD_STREAM')})")
    if n:
        for name, (o, k) in arena.slices.items():
            dd = d[o:o + k]
            c = int(dd.sum())
            if c:
                a, bb = base[o:o + k][dd], runs[rep][o:o + k][dd]
                rel = float(((a - bb).abs() / a.abs().clamp_min(1e-30)).max())
                print(f"   {name}: {c} of {k}, max rel diff {rel:.3e}, first offsets {dd.nonzero().flatten()[:6].tolist()}")
for i in range(2, len(seen_d)):
    dd = seen_d[i] != seen_d[1]
    print(f"type-row gradient handed over by torch, rep {i} vs 1: {int(dd.sum())} differing of {dd.numel()}",
          dd.nonzero().flatten()[:8].tolist())
torch.manual_seed(0)
x = torch.randn(320, 36, 768, device="cuda").bfloat16()
sums = [x.sum((0, 1)) for _ in range(6)]
print("torch bf16 sum over (320, 36) of a fixed tensor, 6 runs: differing elements vs run 0:",
      [int((t != sums[0]).sum()) for t in sums[1:]])
xs = [(x + 0) .sum((0, 1), keepdim=True) for _ in range(6)]
print("   (keepdim):", [int((t != xs[0]).sum()) for t in xs[1:]])
print("gemm fallbacks:", dict(ops.GEMM_FALLBACKS), "rejected candidates:", __import__("vln_bevbert_amd").lib.load().bevbert_gemm_rejected_count())
