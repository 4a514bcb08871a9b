#!/usr/bin/env python3
"""Where does the HOST time of a training step go?  (CPU only, no GPU needed.)

Runs the product model's forward + backward + optimiser step at the bench shapes with every device operation stubbed
out: C-ABI calls return immediately, GEMMs hand back uninitialised tensors, the arena lives in host memory.  Values are
garbage; the Python control flow -- autograd Functions, ctypes argument marshalling, closures of the weight-gradient
stream, torch dispatch of the remaining glue ops -- is exactly the product's, and because the tensors are CPU tensors the
backward runs in the calling thread, where cProfile can see it (on the GPU it runs in autograd's device thread).
What is NOT included: the HIP runtime's own launch cost (~4-5 us per kernel, ~1 000 launches per step).

    python scripts/host_dryrun_profile.py [batch=64] [task=sap]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vln_bevbert_amd import lib, ops, synthetic  # noqa: E402
from vln_bevbert_amd.config import BevBertConfig  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
TASK = sys.argv[2] if len(sys.argv) > 2 else "sap"

# ---- stubs: no device work at all -----------------------------------------------------------------------------------
N_CALLS = {}


def _noop_call(name, *args):
    N_CALLS[name] = N_CALLS.get(name, 0) + 1


# ops.py re-exports the modules below: a stub has to be installed where the CALLERS look the name up
from vln_bevbert_amd import ops_attention, ops_core, ops_gemm, ops_reduce, ops_rowops  # noqa: E402
_MODS = (ops, ops_core, ops_reduce, ops_gemm, ops_rowops, ops_attention)


def _stub(name, value):
    for m in _MODS:
        if hasattr(m, name):
            setattr(m, name, value)


lib.call = _noop_call
_stub("_raw_call", _noop_call)
lib.ptr = lambda t: None if t is None else t.data_ptr()
lib.stream = lambda: 0
_stub("ptr", lib.ptr)
_stub("stream", lib.stream)
_stub("_linear_fwd", lambda x, w, b: torch.empty(x.shape[:-1] + (w.shape[0],), dtype=x.dtype))
_stub("_linear_dgrad", lambda dy2, w, add=None: torch.empty(dy2.shape[0], w.shape[1], dtype=dy2.dtype))
_stub("_linear_wgrad", lambda dy2, x2, S=1, scratch=False: torch.empty(
    (S, dy2.shape[1], x2.shape[1]) if S > 1 else (dy2.shape[1], x2.shape[1]), dtype=dy2.dtype))
ops.RT.scratch.alloc = lambda nbytes, device: 0         # scratch-ring addresses are only ever passed to the (stubbed) C ABI
ops.ReduceQueue.flush = classmethod(lambda cls, device: (cls.jobs.clear(), cls.accum_jobs.clear()))
_stub("_partial_rows", lambda rows: min(512, (rows + 15) // 16))
ops.WgradStream.active = classmethod(lambda cls, device: False)      # closures run inline (their cost is still counted)

from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining  # noqa: E402
from vln_bevbert_amd.train import PretrainTrainer  # noqa: E402

cfg = BevBertConfig()
torch.manual_seed(0)
model = GlocalTextPathCMTPreTraining(cfg)
arena = model.finalize("cpu", torch.bfloat16)
model.train()
model.set_dropout(0.1)
arena.clip_and_step = lambda *a, **k: None          # three launches, negligible host cost
trainer = PretrainTrainer(model, arena)
batch = synthetic.make_batch(cfg, TASK, B, seed=1, sems_as="ids")
batch = {k: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.dtype == torch.float32 and k in ("rgbs",) else v)
         for k, v in batch.items()}


def step():
    trainer.step(TASK, dict(batch))


for _ in range(2):
    step()
N_CALLS.clear()
t0 = time.perf_counter()
n = 5
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
dt = (time.perf_counter() - t0) / n
print(f"dry-run host time: {dt * 1e3:.1f} ms per {TASK} step at batch {B} (with cProfile overhead), "
      f"{sum(N_CALLS.values()) / n:.0f} C-ABI calls per step")
print("C-ABI calls per step:", {k: v // n for k, v in sorted(N_CALLS.items(), key=lambda kv: -kv[1])})
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
print("---- product functions by cumulative time (ms per step, cProfile-inflated) ----")
rows = [(ct / n * 1e3, nc // n, f"{os.path.basename(fn)}:{ln}({name})") for (fn, ln, name), (cc, nc, tt, ct, _) in st.stats.items()
        if "vln_bevbert_amd" in fn]
for ct, nc, label in sorted(rows, reverse=True)[:22]:
    print(f"{ct:8.2f} ms  {nc:5d} calls  {label}")
