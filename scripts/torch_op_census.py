#!/usr/bin/env python3
"""Census of the PyTorch-native device kernels left in one eager training step per task (GPU box): torch.profiler with
shapes, grouped by (op, input shapes), sorted by device time -- the list of what is still NOT a C-ABI launch.
Usage: python scripts/torch_op_census.py [--batch 64] > gpurun_out/torch_op_census.txt"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BEVBERT_GRAPHS", "0")
from vln_bevbert_amd import ops, synthetic  # noqa: E402
from vln_bevbert_amd.config import BevBertConfig  # noqa: E402
from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining  # noqa: E402
from vln_bevbert_amd.static_step import StaticBatch  # noqa: E402
from vln_bevbert_amd.train import PretrainTrainer, load_gemm_tuning  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda", 0)
load_gemm_tuning()
ops.load_gemm_tuning_table()
cfg = BevBertConfig()
torch.manual_seed(0)
model = GlocalTextPathCMTPreTraining(cfg)
arena = model.finalize(dev, torch.bfloat16)
model.train()
model.set_dropout(0.1)
trainer = PretrainTrainer(model, arena)
tasks = ("mlm", "sap", "masksem")
batches = {t: StaticBatch(cfg, t, synthetic.make_batch(cfg, t, a.batch, seed=1000, sems_as="ids"), dev) for t in tasks}
for _ in range(2):
    for t in tasks:
        trainer.step(t, batches[t])
torch.cuda.synchronize()

from torch.profiler import ProfilerActivity, profile  # noqa: E402

for t in tasks:
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        trainer.step(t, batches[t])
        torch.cuda.synchronize()
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0]
    rows.sort(key=lambda e: -e.self_device_time_total)
    tot = sum(e.self_device_time_total for e in rows)
    n = sum(e.count for e in rows)
    print(f"\n===== task {t}: {n} device-launching torch ops, {tot / 1e3:.3f} ms of device time in them")
    rows = [e for e in rows if e.key.startswith("aten::")]
    print(f"      aten ops: {sum(e.count for e in rows)} launches, {sum(e.self_device_time_total for e in rows) / 1e3:.3f} ms")
    for e in rows:
        print(f"{e.count:4d} x {e.self_device_time_total / max(e.count, 1):8.1f} us = {e.self_device_time_total / 1e3:7.3f} ms  "
              f"{e.key:32s} {str(e.input_shapes)[:150]}")
