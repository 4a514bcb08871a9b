#!/usr/bin/env python3
"""One attention shape through the C ABI, kernel by kernel (HIP events around back-to-back launches of each entry):
keep-bit generation, forward, backward.  usage: bench_attn_shape.py [B Lq Lk p [iters]]   (default 64 441 441 0.1 20)
Environment knobs of the library apply (BEVBERT_ATTN_FWD=1 / BEVBERT_ATTN_BWD=1: round-2 kernels)."""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vln_bevbert_amd import ops  # noqa: E402
from vln_bevbert_amd.lib import call, dtype_code, ptr, stream  # noqa: E402


def main():
    a = sys.argv[1:]
    B, Lq, Lk = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (64, 441, 441)
    p = float(a[3]) if len(a) >= 4 else 0.1
    iters = int(a[4]) if len(a) >= 5 else 20
    masked = len(a) >= 6 and a[5] == "mask"
    nh, H, dev = 12, 768, "cuda"
    torch.manual_seed(0)
    q = torch.randn(B, Lq, H, device=dev).bfloat16()
    k = torch.randn(B, Lk, H, device=dev).bfloat16()
    v = torch.randn(B, Lk, H, device=dev).bfloat16()
    do = torch.randn(B, Lq, H, device=dev).bfloat16()
    o = torch.empty_like(q)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lse = torch.empty(B, nh, Lq, dtype=torch.float32, device=dev)
    delta = torch.empty_like(lse)
    km = torch.zeros(B, Lk, device=dev) if masked else None
    bits = ops.attn_drop_bits(B, nh, Lq, Lk, p, 1, 0, dev) if p > 0 else None
    st = ops._strides(q, k, v, o)
    scale = 1.0 / math.sqrt(64)

    def gen():
        call("bevbert_attn_drop_bits", ptr(bits), B, nh, Lq, Lk, p, 1, 0, stream())

    def fwd():
        call("bevbert_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(km), None, st, B, nh, Lq, Lk, 64, scale,
             dtype_code(q), 2, p, 1, 0, ptr(bits), 1, stream())

    trace = torch.zeros(8 * 16 * 8, dtype=torch.int64, device=dev) if os.environ.get("BEVBERT_B2_TRACE") == "1" else None

    def bwd():
        call("bevbert_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk),
             ptr(dv), ptr(trace), ptr(km), None, st, B, nh, Lq, Lk, 64, scale, dtype_code(q), 2, p, 1, 0, ptr(bits), stream())

    def timeit(fn):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3

    fl = 4.0 * B * nh * Lq * Lk * 64
    out = {"B": B, "Lq": Lq, "Lk": Lk, "p": p, "mask": masked,
           "env": {k_: v_ for k_, v_ in os.environ.items() if k_.startswith("BEVBERT_")}}
    if p > 0:
        out["bits_us"] = round(timeit(gen), 1)
    t = timeit(fwd)
    out["fwd_us"], out["fwd_tflops"], out["fwd_frac"] = round(t, 1), round(fl / t / 1e6, 1), round(fl / t / 1e6 / 2500, 4)
    t = timeit(bwd)
    out["bwd_us"], out["bwd_tflops"], out["bwd_frac"] = round(t, 1), round(2.5 * fl / t / 1e6, 1), round(2.5 * fl / t / 1e6 / 2500, 4)
    print(json.dumps(out), flush=True)
    if trace is not None:
        t = trace.cpu().view(8, 16, 8).numpy()
        t0 = int(t[:, 0, 0][t[:, 0, 0] > 0].min())
        print("# workgroup 0: s_memtime stamps relative to the first, per wave (rows) and step; slots 0..4 = step start, "
              "[dQ wave: loads issued | key waves: -], [dQ done | S/dP/softmax done], [tile committed | dK/dV done], barrier passed")
        for w_ in range(8):
            for st_ in range(0, 14):
                row = t[w_, st_]
                print(f"wave {w_} step {st_:2d}: " + " ".join(f"{int(x) - t0:8d}" if x else "       -" for x in row[:8]))


if __name__ == "__main__":
    main()
