#!/usr/bin/env python3
"""cProfile of the host-side enqueue cost of training steps (bench shapes)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vln_bevbert_amd import synthetic  # noqa: E402
from vln_bevbert_amd.config import BevBertConfig  # noqa: E402
from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining  # noqa: E402
from vln_bevbert_amd.train import PretrainTrainer, load_gemm_tuning  # noqa: E402

load_gemm_tuning()
dev = torch.device("cuda:0")
cfg = BevBertConfig()
torch.manual_seed(0)
model = GlocalTextPathCMTPreTraining(cfg)
arena = model.finalize(dev, torch.bfloat16)
model.train(); model.set_dropout(0.1)
tr = PretrainTrainer(model, arena)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bs = {t: synthetic.batch_to(synthetic.make_batch(cfg, t, B, seed=1, sems_as="ids"), dev) for t in ("mlm", "sap")}
for _ in range(3):
    tr.step("sap", bs["sap"]); tr.step("mlm", bs["mlm"])
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(5):
    tr.step("sap", bs["sap"]); tr.step("mlm", bs["mlm"])
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue {(t1 - t0) / 10 * 1e3:.2f} ms/step (profiled), wall {(time.perf_counter() - t0) / 10 * 1e3:.2f}")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
