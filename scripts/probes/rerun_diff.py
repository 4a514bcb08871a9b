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
cfg, B, L = BevBertConfig(), 64, 80
torch.manual_seed(0)
model = GlocalTextPathCMTPreTraining(cfg)
arena = model.finalize("cuda", torch.bfloat16)
model.train()
model.set_dropout(0.1)
b = synthetic.batch_to(synthetic.make_batch(cfg, task, B, seed=2000, txt_len=L, sems_as="ids"), "cuda")
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
