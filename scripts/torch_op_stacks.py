#!/usr/bin/env python3
"""Where the remaining PyTorch-native ops of an eager training step are issued from (GPU box): a TorchDispatchMode logs
every aten op that touches a device tensor (views excluded) with the innermost Python frames inside vln_bevbert_amd/ --
ops with no such frame below backward() are autograd's own (gradient accumulation, view gradients).
Usage: python scripts/torch_op_stacks.py [--residual bf16] [task ...] > gpurun_out/torch_op_stacks.txt"""
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BEVBERT_GRAPHS", "0")
from vln_bevbert_amd import ops, synthetic  # noqa: E402
from vln_bevbert_amd.config import BevBertConfig  # noqa: E402
from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining  # noqa: E402
from vln_bevbert_amd.static_step import StaticBatch  # noqa: E402
from vln_bevbert_amd.train import PretrainTrainer, load_gemm_tuning  # noqa: E402

args = sys.argv[1:]
residual = torch.float32
if "--residual" in args:
    i = args.index("--residual")
    residual = None if args[i + 1] == "bf16" else torch.float32
    del args[i:i + 2]
tasks = args or ["mlm", "sap", "masksem"]
dev = torch.device("cuda", 0)
load_gemm_tuning()
ops.load_gemm_tuning_table()
cfg = BevBertConfig()
torch.manual_seed(0)
model = GlocalTextPathCMTPreTraining(cfg)
arena = model.finalize(dev, torch.bfloat16, residual)
model.train()
model.set_dropout(0.1)
trainer = PretrainTrainer(model, arena)
batches = {t: StaticBatch(cfg, t, synthetic.make_batch(cfg, t, 64, seed=1000, sems_as="ids"), dev) for t in tasks}
for _ in range(2):
    for t in tasks:
        trainer.step(t, batches[t])
torch.cuda.synchronize()

VIEWS = ("view", "reshape", "expand", "slice", "select", "transpose", "permute", "unsqueeze", "squeeze", "detach", "alias",
         "as_strided", "t.default", "split", "narrow", "unbind", "_unsafe_view", "empty", "lift_fresh", "is_", "size", "stride",
         "unfold", "chunk", "numel", "record_stream", "_local_scalar_dense", "item", "set_", "_to_copy")   # (_to_copy handled below)


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = {}

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        flat = [a for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
        outs = [out] if torch.is_tensor(out) else [o for o in (out if isinstance(out, (tuple, list)) else []) if torch.is_tensor(o)]
        on_dev = any(t.is_cuda for t in flat + outs)
        is_view = any(v in name for v in VIEWS if v != "_to_copy") and "_to_copy" not in name
        if on_dev and not is_view:
            frames = [f for f in traceback.extract_stack() if "vln_bevbert_amd" in f.filename]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in frames[-3:][::-1]) or "(autograd engine)"
            shapes = ",".join(str(tuple(t.shape)) for t in flat[:3])
            key = (name.replace("aten.", ""), shapes, where)
            self.seen[key] = self.seen.get(key, 0) + 1
        return out


for t in tasks:
    log = Log()
    with log:
        trainer.step(t, batches[t])
        torch.cuda.synchronize()
    print(f"\n===== task {t}: {sum(log.seen.values())} aten ops on device tensors (views excluded)")
    for (name, shapes, where), n in sorted(log.seen.items(), key=lambda kv: (kv[0][2], kv[0][0])):
        print(f"{n:3d} x {name:28s} {shapes:58s} {where}")
