#!/usr/bin/env python3
"""A/B of weight-gradient formulations through the C ABI (bevbert_gemm_*), per step shape:
  a) split-K strided-batch GEMM -> bf16 partials, then bevbert_accum_partials into the fp32 sink   (2 launches)
  b) one GEMM bf16 x bf16 -> fp32 accumulated straight into the sink (beta = 1)                    (1 launch)
  c) split-K strided-batch GEMM -> fp32 partials + accum_partials
Times are device-side (events around 30 back-to-back iterations)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vln_bevbert_amd import ops  # noqa: E402
from vln_bevbert_amd.lib import call, dtype_code, ptr, stream  # noqa: E402


def t(fn, n=30):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


dev = "cuda"
for (M, N, K) in [(28224, 768, 768), (28224, 2304, 768), (28224, 3072, 768), (28224, 768, 3072), (28224, 1536, 768),
                  (5120, 768, 768), (5120, 2304, 768), (5120, 3072, 768), (5120, 768, 3072), (5120, 1536, 768),
                  (11520, 768, 768), (11520, 2304, 768), (11520, 3072, 768), (11520, 768, 3072), (11520, 768, 512)]:
    dy = torch.randn(M, N, device=dev).bfloat16()
    x = torch.randn(M, K, device=dev).bfloat16()
    sink = torch.zeros(N, K, device=dev)
    fl = 2.0 * M * N * K
    S = ops._split_k(M, N, K)

    def a():
        part = ops._linear_wgrad(dy, x, S)
        call("bevbert_accum_partials", ptr(part), ptr(sink), S, N * K, dtype_code(part), stream())

    def b():
        assert ops._lt_gemm(dy, x, sink, None, N, K, M, 1, 0, N, K, K, accumulate=1)

    def c():
        part = torch.empty(S, N, K, device=dev)
        Ms = M // S
        assert ops._lt_gemm(dy, x, part, None, N, K, Ms, 1, 0, N, K, K, S, Ms * N, Ms * K, N * K)
        call("bevbert_accum_partials", ptr(part), ptr(sink), S, N * K, dtype_code(part), stream())

    row = [f"M={M:5d} N={N:4d} K={K:4d} S={S:2d}"]
    for name, fn in (("splitK+accum", a), ("beta1->fp32", b), ("splitK fp32+accum", c)):
        us = t(fn)
        row.append(f"{name} {us:6.1f}us {fl / us / 1e6:5.0f}TF")
    print("  ".join(row), flush=True)
