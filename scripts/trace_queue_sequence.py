#!/usr/bin/env python3
"""One replayed step of a rocprofv3 --kernel-trace CSV as a time-ordered sequence with the hardware queue of every
dispatch: how the graph executor deals the captured step's kernels onto the queues, how often consecutive kernels change
queue, and how long the chip idles between the end of one kernel and the start of the next when nothing else runs.
Usage: trace_queue_sequence.py <dir with *kernel_trace*.csv> [step from the end = 2] [rows to print = 150]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nprint = int(sys.argv[3]) if len(sys.argv) > 3 else 150
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace*.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
ev.sort()
ends = [e for e in ev if e[2].startswith("adamw_kernel")]
t0, t1 = ends[-back - 1][1], ends[-back][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]


def short(n):
    n = n.replace("void ", "")
    if n.startswith("Cijk") or n.startswith("Custom_Cijk"):
        i = n.find("_MT")
        return "gemm" + n[i:i + 14]
    return n.split("(")[0][:44]


print(f"step of {(t1 - t0) / 1e6:.2f} ms, {len(win)} kernels; queues: "
      + str({q: round(sum(e - s for s, e, _, qq in win if qq == q) / 1e6, 2) for q in sorted({w[3] for w in win})}))
# queue changes between kernels that follow each other in START order
chg = sum(1 for a, b in zip(win, win[1:]) if a[3] != b[3])
print(f"consecutive dispatches on different queues: {chg} of {len(win) - 1}")
# idle time: gaps of the union of intervals
gaps, cur_e = [], None
for s, e, *_ in win:
    if cur_e is not None and s > cur_e:
        gaps.append(s - cur_e)
    cur_e = e if cur_e is None else max(cur_e, e)
print(f"idle gaps: {len(gaps)}, total {sum(gaps) / 1e3:.1f} us, mean {sum(gaps) / max(1, len(gaps)) / 1e3:.2f} us, "
      f"max {max(gaps or [0]) / 1e3:.1f} us")
# per queue: runs of consecutive kernels (in that queue's own order) and the gap between them
for q in sorted({w[3] for w in win}):
    mine = [w for w in win if w[3] == q]
    g = [b[0] - a[1] for a, b in zip(mine, mine[1:])]
    tight = sum(1 for x in g if x < 3000)
    print(f"queue {q}: {len(mine)} kernels, {sum(e - s for s, e, *_ in mine) / 1e6:.2f} ms busy, back-to-back (< 3 us) {tight} of {len(g)}, "
          f"median gap {sorted(g)[len(g) // 2] / 1e3 if g else 0:.1f} us")
print()
print(f"{'t_us':>9s} {'dur_us':>8s} q  kernel")
for s, e, n, q in win[:nprint]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {q}  {short(n)}")
