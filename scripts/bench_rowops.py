#!/usr/bin/env python3
"""The HBM-bound row kernels through the C ABI at the two row counts of the R2R step (64 x 80 text tokens, 64 x 441 BEV
cells): bias+dropout+residual+LayerNorm forward / backward, bias+GELU forward / backward, column sums.  Reports the
average launch time over back-to-back launches and the ALGORITHMIC bytes / time.  usage: bench_rowops.py [iters]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vln_bevbert_amd import ops  # noqa: E402,F401  (loads the library, registers the step salt)
from vln_bevbert_amd.lib import call, ptr, stream  # noqa: E402

BF16, H = 1, 768
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def timeit(fn):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(ITERS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / ITERS * 1e3


def main():
    dev = "cuda"
    ws_floats = 512 * 3 * 3072          # bevbert_colsum_workspace_floats(3 * 3072)
    ws = torch.empty(ws_floats, device=dev)
    for rows in (5120, 28224):
        x = torch.randn(rows, H, device=dev).bfloat16()
        res = torch.randn(rows, H, device=dev).bfloat16()
        y, z, dz, dx = (torch.empty_like(x) for _ in range(4))
        mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
        bias, gamma, beta = torch.randn(H, device=dev), torch.randn(H, device=dev), torch.randn(H, device=dev)
        dy = torch.randn(rows, H, device=dev).bfloat16()
        for p in (0.1, 0.0):
            t = timeit(lambda: call("bevbert_bias_dropout_residual_layernorm_fwd", ptr(x), ptr(bias), ptr(res), ptr(gamma),
                                    ptr(beta), ptr(y), ptr(z), ptr(mean), ptr(rstd), rows, H, 1e-12, BF16, p, 1, 0, stream()))
            nb = rows * H * 2 * 4
            print(json.dumps({"kernel": "ln_fwd", "rows": rows, "p": p, "us": round(t, 2), "GBps": round(nb / t / 1e3, 1)}), flush=True)
            t = timeit(lambda: call("bevbert_layernorm_bwd", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(dz),
                                    ptr(dx) if p > 0 else None, None, None, None, ptr(ws), rows, H, BF16, p, 1, 0, 0, stream()))
            nb = rows * H * 2 * (4 if p > 0 else 3)
            print(json.dumps({"kernel": "ln_bwd", "rows": rows, "p": p, "us": round(t, 2), "GBps": round(nb / t / 1e3, 1)}), flush=True)
        C = 3072
        xi = torch.randn(rows, C, device=dev).bfloat16()
        yi, dyi, dxi = torch.empty_like(xi), torch.randn(rows, C, device=dev).bfloat16(), torch.empty_like(xi)
        bi = torch.randn(C, device=dev)
        t = timeit(lambda: call("bevbert_bias_gelu_fwd", ptr(xi), ptr(bi), ptr(yi), rows, C, BF16, stream()))
        print(json.dumps({"kernel": "gelu_fwd", "rows": rows, "us": round(t, 2), "GBps": round(rows * C * 4 / t / 1e3, 1)}), flush=True)
        t = timeit(lambda: call("bevbert_bias_gelu_bwd", ptr(dyi), ptr(xi), ptr(bi), ptr(dxi), None, ptr(ws), rows, C, BF16, 0, stream()))
        print(json.dumps({"kernel": "gelu_bwd", "rows": rows, "us": round(t, 2), "GBps": round(rows * C * 6 / t / 1e3, 1)}), flush=True)
        for Cc in (768, 2304):
            d = torch.randn(rows, Cc, device=dev).bfloat16()
            t = timeit(lambda: call("bevbert_colsum_partials", ptr(d), ptr(ws), rows, Cc, BF16, stream()))
            print(json.dumps({"kernel": "colsum_partials", "rows": rows, "C": Cc, "us": round(t, 2),
                              "GBps": round(rows * Cc * 2 / t / 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
