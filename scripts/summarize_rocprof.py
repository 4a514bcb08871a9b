#!/usr/bin/env python3
"""Condense rocprofv3 --kernel-trace --stats output into a short per-kernel table (top 40 by total time),
grouping the hand-written kernels (libbevbert_hip.so) apart from library GEMMs (hipBLASLt/rocBLAS/Tensile) and
PyTorch elementwise kernels."""
import csv
import glob
import os
import sys

d = sys.argv[1]
files = glob.glob(os.path.join(d, "*kernel_stats*.csv"))
if not files:
    print("no kernel_stats csv found in", d)
    sys.exit(0)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)


def cls(name):
    n = name.lower()
    if any(k in n for k in ("attn_", "ln_fwd", "ln_bwd", "bev_", "colwise", "colsum", "bias_gelu", "gather_wsum",
                            "adamw", "sumsq", "clip_coef", "cast_f32", "keep_mask")):
        return "custom"
    if any(k in n for k in ("cijk", "gemm", "tensile", "hipblaslt", "rocblas")):
        return "gemm"
    if "nccl" in n or "rccl" in n:
        return "rccl"
    return "torch"


key_t = "TotalDurationNs" if "TotalDurationNs" in rows[0] else [k for k in rows[0] if "total" in k.lower()][0]
key_c = "Calls" if "Calls" in rows[0] else [k for k in rows[0] if "call" in k.lower()][0]
key_a = "AverageNs" if "AverageNs" in rows[0] else [k for k in rows[0] if "average" in k.lower()][0]
tot = sum(float(r[key_t]) for r in rows)
by = {}
for r in rows:
    c = cls(r["Name"])
    by[c] = by.get(c, 0.0) + float(r[key_t])
print(f"total kernel time {tot / 1e6:.2f} ms over {len(rows)} distinct kernels")
for c, t in sorted(by.items(), key=lambda kv: -kv[1]):
    print(f"  {c:7s} {t / 1e6:10.2f} ms  {100 * t / tot:5.1f} %")
print()
print(f"{'class':7s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'%':>6s}  name")
for r in sorted(rows, key=lambda r: -float(r[key_t]))[:40]:
    print(f"{cls(r['Name']):7s} {int(float(r[key_c])):7d} {float(r[key_t]) / 1e6:10.3f} {float(r[key_a]) / 1e3:10.2f} "
          f"{100 * float(r[key_t]) / tot:6.2f}  {r['Name'][:110]}")
