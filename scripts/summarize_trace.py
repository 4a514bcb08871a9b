#!/usr/bin/env python3
"""Per-step view of a rocprofv3 --kernel-trace CSV: keeps only the last K training steps (delimited by adamw_kernel
launches, so warm-up and the first-use GEMM autotuning are excluded) and reports wall time, GPU busy time (union of
kernel intervals), the sum of kernel durations (> busy when streams overlap) and the per-kernel table.
Usage: summarize_trace.py <dir with *kernel_trace*.csv> [K=11]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 11
files = glob.glob(os.path.join(d, "**", "*kernel_trace*.csv"), recursive=True)
if not files:
    print("no kernel_trace csv under", d)
    sys.exit(0)
ev = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            wg = max(1, int(r.get("Workgroup_Size_X", 1) or 1))
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
                       int(r.get("Grid_Size_X", 0) or 0) // wg))
ev.sort()
ends = [e for e in ev if e[2].startswith("adamw_kernel")]
if len(ends) < K + 1:
    print(f"only {len(ends)} steps in the trace")
    sys.exit(0)
t0, t1 = ends[-K - 1][1], ends[-1][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
wall = (t1 - t0) / 1e6
busy, cur_s, cur_e = 0, None, None
for s, e, *_ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    busy += cur_e - cur_s
tot = sum(e - s for s, e, *_ in win)


def cls(name):
    n = name.lower()
    if any(k in n for k in ("attn_", "ln_fwd", "ln_bwd", "bev_", "colwise", "colsum", "bias_gelu", "gather_wsum",
                            "adamw", "sumsq", "clip_coef", "cast_f32", "keep_mask", "accum_partials", "dropout_add",
                            "embedding_grad", "multi_accum", "multi_finalize", "sap_loss", "ce_fwd", "ce_bwd", "gm_",
                            "colsum_finalize", "ln_res32", "smallk_", "rows_gather", "rows_scatter", "graph_bias", "weighted_mean", "colsum_any")):
        return "custom"
    if any(k in n for k in ("cijk", "gemm", "tensile", "hipblaslt", "rocblas")):
        return "gemm"
    if "nccl" in n or "rccl" in n:
        return "rccl"
    if "rocclr_fillbuffer" in n or "rocclr_copybuffer" in n:
        return "memop"
    return "torch"


print(f"last {K} steps: wall {wall:.2f} ms ({wall / K:.2f} ms/step), GPU busy {busy / 1e6 / K:.2f} ms/step "
      f"({100 * busy / 1e6 / wall:.1f} % of wall), sum of kernel durations {tot / 1e6 / K:.2f} ms/step, "
      f"{len(win) / K:.0f} launches/step")
by_q = {}
for s, e, _, q, _ in win:
    by_q[q] = by_q.get(q, 0) + (e - s)
print("per HW queue (ms/step):", {q: round(t / 1e6 / K, 2) for q, t in sorted(by_q.items())})
by_c, by_k = {}, {}
for s, e, n, _, _ in win:
    by_c[cls(n)] = by_c.get(cls(n), 0) + (e - s)
    c = by_k.setdefault(n, [0, 0])
    c[0] += 1
    c[1] += e - s
for c, t in sorted(by_c.items(), key=lambda kv: -kv[1]):
    print(f"  {c:7s} {t / 1e6 / K:8.3f} ms/step  {100 * t / tot:5.1f} %")
print()
print(f"{'class':7s} {'calls/step':>10s} {'ms/step':>9s} {'avg_us':>9s} {'%':>6s}  name")
for n, (c, t) in sorted(by_k.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{cls(n):7s} {c / K:10.1f} {t / 1e6 / K:9.3f} {t / 1e3 / c:9.2f} {100 * t / tot:6.2f}  {n[:110]}")

# every PyTorch eager kernel left in the step (<= 10 at::native launches per step)
print()
tk = [(n, c, t) for n, (c, t) in by_k.items() if cls(n) == "torch"]
print(f"PyTorch eager kernels in the step: {sum(c for _, c, _ in tk) / K:.1f} launches/step, {sum(t for _, _, t in tk) / 1e6 / K:.3f} ms/step")
for n, c, t in sorted(tk, key=lambda x: -x[2]):
    print(f"  {c / K:6.1f} calls/step {t / 1e3 / c:8.2f} us avg {t / 1e6 / K:7.3f} ms/step  {n[:150]}")

# The hand-written kernels per PROBLEM: one kernel symbol serves several shapes of a step (attn_bwd2_kernel: the 441 x 441
# BEV self-attention and the 80 x 441 text <- BEV cross-attention share a symbol AND a launch grid).  The kernel trace
# carries no kernel arguments, so launches are keyed by (symbol, grid) and then split into duration clusters (sorted
# durations, a new cluster wherever the next one is more than 1.35 x the previous): each line is one problem size, its
# in-step duration is the figure to hold against bench.py's roofline (isolated launches between HIP events).
print()
print("hand-written kernels by (symbol, launch grid, duration cluster): calls/step, avg / min / max us in the step")
by_g = {}
for s, e, n, _, g in win:
    if cls(n) == "custom":
        by_g.setdefault((n.split("(")[0], g), []).append((e - s) / 1e3)
rows = []
for (n, g), ds in by_g.items():
    ds.sort()
    cl = [[ds[0]]]
    for d_ in ds[1:]:
        if d_ > 1.35 * cl[-1][-1] and d_ - cl[-1][-1] > 3.0:
            cl.append([])
        cl[-1].append(d_)
    for c in cl:
        rows.append((sum(c), n, g, len(c), c))
for t, n, g, c, ds in sorted(rows, reverse=True)[:60]:
    print(f"  {c / K:6.1f} calls/step {t / c:9.2f} us avg {ds[0]:9.2f} min {ds[-1]:9.2f} max  {t / K / 1e3:7.3f} ms/step  grid {g:6d}  {n[:70]}")
