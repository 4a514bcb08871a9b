#!/usr/bin/env python3
"""Noise floor of repeated 4-step training runs (tiny config, bf16) with / without the one-rank RCCL exchange."""
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vln_bevbert_amd import synthetic, weights  # noqa: E402
from vln_bevbert_amd.config import BevBertConfig  # noqa: E402
from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining  # noqa: E402
from vln_bevbert_amd.train import PretrainTrainer  # noqa: E402

DEV = torch.device("cuda", 0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
cfg = BevBertConfig.tiny(num_l_layers=1, num_x_layers=1, vocab_size=400)
dtype = torch.float32 if "fp32" in sys.argv else torch.bfloat16


def run(force, overlap=True, steps=4):
    model = GlocalTextPathCMTPreTraining(cfg)
    model.load_state_dict(weights.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
    model.tie_weights()
    arena = model.finalize(DEV, dtype)
    model.train()
    model.set_dropout(0.1)
    tr = PretrainTrainer(model, arena, warmup_steps=2, num_train_steps=20, force_collectives=force, overlap=overlap)
    grads = []
    for i, task in enumerate(("sap", "mlm", "masksem", "sap")[:steps]):
        tr.step(task, synthetic.batch_to(synthetic.make_batch(cfg, task, 2, seed=80 + i, ragged=True), DEV))
        grads.append(arena.grads.clone())
    torch.cuda.synchronize()
    return arena.params.clone(), grads, arena


def diff(a, b):
    d = (a - b).abs()
    return f"max {float(d.max()):.3e} mean {float(d.mean()):.3e} nnz {int((d > 0).sum())}/{d.numel()}"


runs = {k: run(*v) for k, v in {"F1": (False,), "F2": (False,), "T1": (True,), "T2": (True,), "Tno": (True, False)}.items()}
for a, b in (("F1", "F2"), ("T1", "T2"), ("F1", "T1"), ("F1", "Tno")):
    print(a, b, "params:", diff(runs[a][0], runs[b][0]))
    for i, (ga, gb) in enumerate(zip(runs[a][1], runs[b][1])):
        print("    step", i + 1, "grads:", diff(ga, gb))
# which parameters differ most between F1 and T1 after step 1
arena = runs["F1"][2]
d = (runs["F1"][1][0] - runs["T1"][1][0]).abs()
rows = []
for name, (lo, k) in arena.slices.items():
    rows.append((float(d[lo:lo + k].max()), name, lo))
rows.sort(reverse=True)
print("top gradient diffs after step 1 (F1 vs T1):")
for r in rows[:12]:
    print("   ", r)
dist.destroy_process_group()
