#!/usr/bin/env python3
"""Offline hipBLASLt/rocBLAS solution tuning for the GEMM shapes of the R2R pre-training step (PyTorch TunableOp).

The Linear layers stay on the vendor BLAS by design (north star); which of its kernels runs for a given shape is a
library heuristic.  This script runs a few training steps with TunableOp recording + tuning on and writes
``tunableop_results.csv`` (shape -> solution index), which bench.py / the trainer load read-only.  The file is only
valid for the same PyTorch / ROCm / hipBLASLt build and GPU arch (TunableOp validates that itself and ignores it
otherwise)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "vln_bevbert_amd", "tunableop_results.csv")
os.environ.setdefault("PYTORCH_TUNABLEOP_ROCBLAS_ENABLED", "0")

import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402

from vln_bevbert_amd import synthetic  # noqa: E402
from vln_bevbert_amd.config import BevBertConfig  # noqa: E402
from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining  # noqa: E402
from vln_bevbert_amd.train import PretrainTrainer  # noqa: E402


def main():
    budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
    t0 = time.time()
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(30)          # ms per candidate
    tunable.set_max_tuning_iterations(10)
    tunable.set_filename(OUT)
    dev = torch.device("cuda:0")
    cfg = BevBertConfig()
    torch.manual_seed(0)
    model = GlocalTextPathCMTPreTraining(cfg)
    arena = model.finalize(dev, torch.bfloat16)
    model.train()
    model.set_dropout(0.1)
    trainer = PretrainTrainer(model, arena)
    for task in ("sap", "mlm", "masksem"):
        b = synthetic.batch_to(synthetic.make_batch(cfg, task, 64, seed=1000, sems_as="ids"), dev)
        trainer.step(task, b)
        torch.cuda.synchronize()
        print(f"[tune +{time.time() - t0:6.1f}s] {task}: {len(tunable.get_results())} tuned shapes", flush=True)
        if time.time() - t0 > budget_s:
            print("budget exhausted", flush=True)
            break
    # the table is flushed to OUT when the process exits (write_file_on_exit)


if __name__ == "__main__":
    main()
