#!/usr/bin/env python3
"""A/B of weight-gradient GEMM formulations dW(N x Kin) = dy^T(N x M) @ x(M x Kin) for the step's shapes:
plain mm vs split-K as a batched GEMM over S chunks of M (+ partial sum)."""
import torch

def t(fn, n=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

dev = "cuda"
for (M, N, K) in [(28224, 768, 768), (28224, 2304, 768), (28224, 3072, 768), (28224, 768, 3072), (28224, 1536, 768),
                  (5120, 768, 768), (5120, 2304, 768), (5120, 3072, 768), (5120, 768, 3072), (5120, 1536, 768),
                  (11520, 768, 768), (11520, 2304, 768), (11520, 3072, 768), (11520, 768, 3072), (11520, 768, 512)]:
    dy = torch.randn(M, N, device=dev).bfloat16()
    x = torch.randn(M, K, device=dev).bfloat16()
    fl = 2.0 * M * N * K
    row = [f"M={M:5d} N={N:4d} K={K:4d}"]
    us = t(lambda: dy.t().mm(x))
    row.append(f"mm {us:7.1f}us {fl / us / 1e6:6.0f}TF")
    for S in (2, 4, 8, 16, 32):
        if M % S: continue
        f = lambda: torch.bmm(dy.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K)).sum(0)
        us = t(f)
        row.append(f"S{S} {us:6.1f}us {fl / us / 1e6:5.0f}TF")
    print("  ".join(row), flush=True)
