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


MARK = os.environ.get("ROWOPS_MARK") == "1"      # PMC passes: a marker launch (bevbert_cast_f32 over 65 536 (idx + 1) elements)
_mark_idx = [0]                                   # in front of every record, so that the counter rows between two markers
_mark_src = None                                  # can be attributed to one record (scripts/gpu_pmc_all.sh)


def _marker():
    global _mark_src
    if not MARK:
        return
    if _mark_src is None:
        _mark_src = (torch.zeros(65536 * 64, device="cuda"), torch.empty(65536 * 64, dtype=torch.bfloat16, device="cuda"))
    _mark_idx[0] += 1
    call("bevbert_cast_f32", ptr(_mark_src[0]), ptr(_mark_src[1]), 65536 * _mark_idx[0], BF16, stream())


def timeit(fn):
    _marker()
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
    ws_floats = int(__import__("vln_bevbert_amd.lib", fromlist=["load"]).load().bevbert_colsum_workspace_floats(3 * 3072))
    ws = torch.empty(ws_floats, device=dev)
    for rows in [int(r) for r in os.environ.get("ROWOPS_ROWS", "5120,28224").split(",")]:
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
            print(json.dumps({"kernel": "ln_fwd", "rows": rows, "p": p, "us": round(t, 2), "bytes": nb, "mark": _mark_idx[0], "GBps": round(nb / t / 1e3, 1)}), flush=True)
            t = timeit(lambda: call("bevbert_layernorm_bwd", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(dz),
                                    ptr(dx) if p > 0 else None, None, None, None, ptr(ws), rows, H, BF16, p, 1, 0, 0, stream()))
            nb = rows * H * 2 * (4 if p > 0 else 3)
            print(json.dumps({"kernel": "ln_bwd", "rows": rows, "p": p, "us": round(t, 2), "bytes": nb, "mark": _mark_idx[0], "GBps": round(nb / t / 1e3, 1)}), flush=True)
        # the fp32-residual-stream pair (bench default since round 6): x bf16 + residual fp32 -> y bf16, y fp32, z fp32;
        # backward dy bf16 + dy fp32 + z fp32 -> dz fp32 + dx bf16
        r32 = torch.randn(rows, H, device=dev)
        y32, z32, dy32, dz32 = (torch.empty(rows, H, device=dev) for _ in range(4))
        for p in (0.1, 0.0):
            t = timeit(lambda: call("bevbert_layernorm_res32_fwd", ptr(x), ptr(bias), ptr(r32), 0, ptr(gamma), ptr(beta), ptr(y),
                                    ptr(y32), ptr(z32), ptr(mean), ptr(rstd), rows, H, 1e-12, p, 1, 0, stream()))
            nb = rows * H * (2 + 4 + 2 + 4 + 4)
            print(json.dumps({"kernel": "ln_res32_fwd", "rows": rows, "p": p, "us": round(t, 2), "bytes": nb, "mark": _mark_idx[0], "GBps": round(nb / t / 1e3, 1)}), flush=True)
            t = timeit(lambda: call("bevbert_layernorm_res32_bwd", ptr(dy), ptr(dy32), ptr(z32), ptr(mean), ptr(rstd), ptr(gamma),
                                    ptr(dz32), ptr(dx), None, None, None, ptr(ws), rows, H, p, 1, 0, 0, 0, stream()))
            nb = rows * H * (2 + 4 + 4 + 4 + 2)
            print(json.dumps({"kernel": "ln_res32_bwd", "rows": rows, "p": p, "us": round(t, 2), "bytes": nb, "mark": _mark_idx[0], "GBps": round(nb / t / 1e3, 1)}), flush=True)
        C = 3072
        xi = torch.randn(rows, C, device=dev).bfloat16()
        yi, dyi, dxi = torch.empty_like(xi), torch.randn(rows, C, device=dev).bfloat16(), torch.empty_like(xi)
        bi = torch.randn(C, device=dev)
        t = timeit(lambda: call("bevbert_bias_gelu_fwd", ptr(xi), ptr(bi), ptr(yi), rows, C, BF16, stream()))
        print(json.dumps({"kernel": "gelu_fwd", "rows": rows, "us": round(t, 2), "bytes": rows * C * 4, "mark": _mark_idx[0], "GBps": round(rows * C * 4 / t / 1e3, 1)}), flush=True)
        t = timeit(lambda: call("bevbert_bias_gelu_bwd", ptr(dyi), ptr(xi), ptr(bi), ptr(dxi), None, ptr(ws), rows, C, BF16, 0, stream()))
        print(json.dumps({"kernel": "gelu_bwd", "rows": rows, "us": round(t, 2), "bytes": rows * C * 6, "mark": _mark_idx[0], "GBps": round(rows * C * 6 / t / 1e3, 1)}), flush=True)
        # the same traffic with a trivial activation gradient: what the erf-GELU arithmetic costs on top of the memory pipeline
        t = timeit(lambda: call("bevbert_bias_relu_bwd", ptr(dyi), ptr(xi), ptr(bi), ptr(dxi), None, ptr(ws), rows, C, BF16, 0, stream()))
        print(json.dumps({"kernel": "relu_bwd", "rows": rows, "us": round(t, 2), "bytes": rows * C * 6, "mark": _mark_idx[0], "GBps": round(rows * C * 6 / t / 1e3, 1)}), flush=True)
        t = timeit(lambda: call("bevbert_bias_relu_fwd", ptr(xi), ptr(bi), ptr(yi), rows, C, BF16, stream()))
        print(json.dumps({"kernel": "relu_fwd", "rows": rows, "us": round(t, 2), "bytes": rows * C * 4, "mark": _mark_idx[0], "GBps": round(rows * C * 4 / t / 1e3, 1)}), flush=True)
        for Cc in (768, 2304):
            d = torch.randn(rows, Cc, device=dev).bfloat16()
            t = timeit(lambda: call("bevbert_colsum_partials", ptr(d), ptr(ws), rows, Cc, BF16, stream()))
            print(json.dumps({"kernel": "colsum_partials", "rows": rows, "C": Cc, "us": round(t, 2), "bytes": rows * Cc * 2,
                              "mark": _mark_idx[0], "GBps": round(rows * Cc * 2 / t / 1e3, 1)}), flush=True)


def extra():
    """The other HBM-bound entries of the step at full size: grid-feature splat (B = 64 samples x 2 352 points x 768, fp16
    store rows in, bf16 BEV out), global gradient norm + AdamW over the 238.8 M-element arena, the batched split-K fold
    (bevbert_multi_accum) on a table shaped like a step's.  Same output records; `bytes` = algorithmic bytes."""
    import numpy as np
    dev = "cuda"
    # ---- splat
    B, P, K, C = 64, 2352, 441, 768
    g = torch.Generator(device="cpu").manual_seed(0)
    feat = torch.randn(B, P, C, generator=g).half().to(dev)
    cells = torch.randint(0, K, (B, P), generator=g)
    order = torch.argsort(cells, dim=1, stable=True).int().to(dev)
    counts = torch.stack([torch.bincount(c_, minlength=K) for c_ in cells])
    cell_start = torch.cat([torch.zeros(B, 1, dtype=torch.long), counts.cumsum(1)], 1).int().to(dev)
    sem = torch.randint(0, 40, (B, P), generator=g).to(torch.uint8).to(dev)
    t = timeit(lambda: ops.bev_splat_mean(feat, order, cell_start, K, out_dtype=torch.bfloat16, sems=sem))
    nb = B * (P * C * 2 + P * 4 + K * C * 2 + P * 1 + K * 41)
    print(json.dumps({"kernel": "bev_splat_mean", "rows": B * P, "us": round(t, 2), "bytes": nb, "mark": _mark_idx[0], "GBps": round(nb / t / 1e3, 1)}), flush=True)
    del feat
    # ---- optimiser
    n = 238_800_000 // 1024 * 1024
    params, grads = torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-3
    m1, m2 = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=dev)
    flags = torch.full((n // 1024,), 3, dtype=torch.uint8, device=dev)
    steps = torch.zeros(n // 1024, dtype=torch.int32, device=dev)
    part, scal = torch.empty(1024, device=dev), torch.empty(2, device=dev)
    t = timeit(lambda: call("bevbert_grad_norm_clip", ptr(grads), n, 1.0, 40.0, ptr(part), ptr(scal), stream()))
    print(json.dumps({"kernel": "grad_norm_clip", "rows": n, "us": round(t, 2), "bytes": n * 4, "mark": _mark_idx[0], "GBps": round(n * 4 / t / 1e3, 1)}), flush=True)
    t = timeit(lambda: call("bevbert_adamw_step", ptr(params), ptr(grads), ptr(m1), ptr(m2), ptr(shadow), ptr(flags), ptr(steps),
                            n, ptr(scal[1:]), None, 5e-5, 0.9, 0.98, 1e-6, 0.01, stream()))
    print(json.dumps({"kernel": "adamw_step", "rows": n, "us": round(t, 2), "bytes": n * 30, "mark": _mark_idx[0], "GBps": round(n * 30 / t / 1e3, 1)}), flush=True)
    # ---- batched split-K fold: 100 products of 768 x 768 .. 3072 x 768 with S = 8 bf16 partial slices each
    jobs, keep = [], []
    for i in range(100):
        nn = 768 * (768 if i % 3 else 3072)
        pt = torch.randn(8, nn, device=dev).bfloat16()
        keep.append(pt)
        jobs.append((pt.data_ptr(), grads.data_ptr() + 4 * ((i * 3072 * 768) % (n - nn)) // 16 * 16, 8, nn, BF16))
    table, cnt = ops.ReduceQueue._build_accum(tuple(jobs), torch.device(dev))
    nb = ops.ReduceQueue.table_bytes[table.data_ptr()]
    t = timeit(lambda: call("bevbert_multi_accum", table.data_ptr(), cnt, stream()))
    print(json.dumps({"kernel": "multi_accum", "rows": cnt, "us": round(t, 2), "bytes": nb, "mark": _mark_idx[0], "GBps": round(nb / t / 1e3, 1)}), flush=True)
    # ---- LayerNorm backward with the residual-stream addend (panorama encoder), word-embedding gradient
    rows = 11520
    dy = torch.randn(rows, H, device=dev).bfloat16()
    z, dz, add = torch.randn_like(dy), torch.empty_like(dy), torch.randn_like(dy)
    mean, rstd, gamma = torch.zeros(rows, device=dev), torch.ones(rows, device=dev), torch.ones(H, device=dev)
    ws = torch.empty(512 * 3 * 3072, device=dev)
    t = timeit(lambda: call("bevbert_layernorm_bwd_add", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(dz), None, ptr(add),
                            None, None, None, ptr(ws), rows, H, BF16, 0.0, 1, 0, 0, stream()))
    print(json.dumps({"kernel": "ln_bwd_add", "rows": rows, "us": round(t, 2), "bytes": rows * H * 2 * 4, "mark": _mark_idx[0], "GBps": round(rows * H * 8 / t / 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
    if os.environ.get("ROWOPS_EXTRA", "1") == "1":
        extra()
