#!/usr/bin/env python3
"""Micro-benchmark of the attention C-ABI entry points at the shapes of one R2R pre-training step (B = 64).
Prints one line per (shape, dropout) with fwd / bwd time and algorithmic TFLOP/s.  Tile-shape knobs are read from the
environment by the library (BEVBERT_FWD_QT / BEVBERT_DQ_QT / BEVBERT_DKV_KT), so run one process per variant."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vln_bevbert_amd import ops  # noqa: E402

SHAPES = [("bev self", 64, 441, 441, False), ("bev<-txt", 64, 441, 80, True), ("txt<-bev", 64, 80, 441, False),
          ("txt self", 64, 80, 80, True), ("pano", 320, 36, 36, True), ("gmap", 64, 17, 17, True)]


def timeit(fn, n=20):
    n = 3 if (len(sys.argv) > 1 and sys.argv[1] == 'one') else n
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3   # us


def main():
    dev = "cuda"
    tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("BEVBERT_")) or "default"
    one = len(sys.argv) > 1 and sys.argv[1] == "one"
    for name, B, Lq, Lk, masked in (SHAPES[:1] if one else SHAPES):
        for p in ((0.1,) if one else (0.0, 0.1)):
            torch.manual_seed(0)
            q = torch.randn(B, Lq, 768, device=dev).bfloat16().requires_grad_(True)
            k = torch.randn(B, Lk, 768, device=dev).bfloat16().requires_grad_(True)
            v = torch.randn(B, Lk, 768, device=dev).bfloat16().requires_grad_(True)
            km = torch.zeros(B, Lk, device=dev) if masked else None
            do = torch.randn(B, Lq, 768, device=dev).bfloat16()
            for impl, iname in ((2, "single-pass bwd"), (3, "two-kernel bwd")):
                fwd = lambda: ops._Attention.apply("sep", q, k, v, km, None, 12, p, impl)
                t_f = timeit(lambda: fwd())
                o = fwd()
                t_b = timeit(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True))
                fl = 4.0 * B * 12 * Lq * Lk * 64
                print(json.dumps({"variant": tag, "impl": iname, "shape": name, "B": B, "Lq": Lq, "Lk": Lk, "p": p,
                                  "fwd_us": round(t_f, 1), "bwd_us": round(t_b, 1),
                                  "fwd_tflops": round(fl / t_f / 1e6, 1),
                                  "bwd_tflops": round(2.5 * fl / t_b / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
