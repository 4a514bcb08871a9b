#!/usr/bin/env python3
"""The short attention shapes of a pre-training step through the C ABI, called the way the model calls them (packed QKV
strides, keep bits hashed inline by the forward and left for the backward when the score matrix is small), timed with HIP
events around back-to-back launches.  One JSON line per shape: forward / backward microseconds, the kernel the library
dispatched to (bevbert_attn_last_path), algorithmic bytes and the HBM-time floor at 8 TB/s.

usage: bench_attn_short.py [--p 0.1] [--iters 50] [--shapes 64x80x80,320x36x36,...] [--mask]
Environment knobs of the library apply (BEVBERT_ATTN_SHORT=0: the kernels of rounds 2-5)."""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vln_bevbert_amd import lib, ops  # noqa: E402
from vln_bevbert_amd.lib import call, dtype_code, ptr, stream  # noqa: E402

# (B, Lq, Lk): text self-attention, panorama encoder (B x T panoramas), global map self / map<-text / text<-map, BEV<-text
DEFAULT = "64x80x80,320x36x36,64x20x20,64x20x80,64x80x20,64x441x80,64x80x441"


def bench(B, Lq, Lk, p, iters, masked):
    nh, H, dev = 12, 768, "cuda"
    torch.manual_seed(0)
    self_attn = Lq == Lk
    if self_attn:                                    # packed QKV, as the fused projection GEMM leaves it
        qkv = torch.randn(B, Lq, 3 * H, device=dev).bfloat16()
        q, k, v = qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[..., :H], dqkv[..., H:2 * H], dqkv[..., 2 * H:]
    else:                                            # cross attention: q alone, K | V packed
        q = torch.randn(B, Lq, H, device=dev).bfloat16()
        kv = torch.randn(B, Lk, 2 * H, device=dev).bfloat16()
        k, v = kv[..., :H], kv[..., H:]
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        dk, dv = dkv[..., :H], dkv[..., H:]
    do = torch.randn(B, Lq, H, device=dev).bfloat16()
    o = torch.empty(B, Lq, H, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, nh, Lq, dtype=torch.float32, device=dev)
    delta = torch.empty_like(lse)
    km = None
    if masked:
        lens = torch.randint(max(1, Lk // 2), Lk + 1, (B,), device=dev)
        km = torch.where(torch.arange(Lk, device=dev)[None] < lens[:, None], 0.0, -10000.0).float().contiguous()
    st = ops._strides(q, k, v, o)
    scale = 1.0 / math.sqrt(64)
    big = Lq * Lk >= 32768 or Lk > 256               # ops_attention._Attention.forward's rule
    bits = None
    if p > 0:
        bits = ops.attn_drop_bits(B, nh, Lq, Lk, p, 1, 0, dev) if big else \
            torch.empty(ops._drop_bits_words(B, nh, Lq, Lk), dtype=torch.int64, device=dev)

    def fwd():
        call("bevbert_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(km), None, st, B, nh, Lq, Lk, 64, scale,
             dtype_code(q), 0, p, 1, 0, ptr(bits), 1 if big else 0, stream())

    def bwd():
        call("bevbert_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk),
             ptr(dv), None, ptr(km), None, st, B, nh, Lq, Lk, 64, scale, dtype_code(q), 0, p, 1, 0, ptr(bits), stream())

    def timeit(fn):
        for _ in range(5):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3

    L = lib.load()
    tf = timeit(fwd)
    pf = L.bevbert_attn_last_path(0).decode()
    tb = timeit(bwd)
    pb = L.bevbert_attn_last_path(1).decode()
    fl = 4.0 * B * nh * Lq * Lk * 64
    by_f = (2 * Lq + 2 * Lk) * H * B * 2
    by_b = (4 * Lq + 4 * Lk) * H * B * 2
    return {"B": B, "Lq": Lq, "Lk": Lk, "p": p, "mask": masked, "fwd_us": round(tf, 2), "fwd_kernel": pf,
            "bwd_us": round(tb, 2), "bwd_kernel": pb, "fwd_MB": round(by_f / 1e6, 1), "bwd_MB": round(by_b / 1e6, 1),
            "fwd_hbm_floor_us": round(by_f / 8e6, 2), "bwd_hbm_floor_us": round(by_b / 8e6, 2),
            "fwd_hbm_frac": round(by_f / 8e6 / tf, 3), "bwd_hbm_frac": round(by_b / 8e6 / tb, 3),
            "fwd_tflops": round(fl / tf / 1e6, 1), "bwd_tflops": round(2.5 * fl / tb / 1e6, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--p", type=float, default=0.1)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--shapes", default=DEFAULT)
    ap.add_argument("--mask", action="store_true")
    a = ap.parse_args()
    env = {k: v for k, v in os.environ.items() if k.startswith("BEVBERT_")}
    for sh in a.shapes.split(","):
        B, Lq, Lk = (int(x) for x in sh.split("x"))
        d = bench(B, Lq, Lk, a.p, a.iters, a.mask)
        d["env"] = env
        print(json.dumps(d), flush=True)


if __name__ == "__main__":
    main()
