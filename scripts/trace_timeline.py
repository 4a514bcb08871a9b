#!/usr/bin/env python3
"""Timeline of ONE training step of a multi-queue run (rocprofv3 --kernel-trace CSV): per hardware queue the busy time,
the first / last kernel, and -- for the queue with the most kernel time (the main stream) -- its idle gaps above a
threshold together with what the other queues were executing meanwhile.  Tells whether the step's critical path is the
main stream's own kernels or its waits for the side streams.
Usage: trace_timeline.py <dir with *kernel_trace*.csv> [step_from_end=2] [gap_us=40]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gap_us = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace*.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-70:], r.get("Queue_Id", "?")))
ev.sort()
ends = [e for e in ev if e[2].endswith("adamw_kernel")]
t0, t1 = ends[-back - 1][1], ends[-back][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print(f"step window {(t1 - t0) / 1e3:.1f} us, {len(win)} kernels")
qs = {}
for e in win:
    qs.setdefault(e[3], []).append(e)
main = max(qs, key=lambda q: sum(e[1] - e[0] for e in qs[q]))
for q, es in sorted(qs.items()):
    busy = sum(e[1] - e[0] for e in es)
    print(f"queue {q}{' (main)' if q == main else ''}: {len(es):4d} kernels, busy {busy / 1e3:8.1f} us, first at +{(es[0][0] - t0) / 1e3:8.1f} "
          f"({es[0][2][:40]}), last ends at +{(es[-1][1] - t0) / 1e3:8.1f} ({es[-1][2][:40]})")
print(f"\nidle gaps of the main queue above {gap_us:.0f} us (time into the step, gap, next main kernel | busy time of the other queues inside the gap)")
es = qs[main]
tot_gap = 0.0
prev_end = t0
for e in es + [(t1, t1, "<end of step>", main)]:
    g = (e[0] - prev_end) / 1e3
    if g >= gap_us:
        tot_gap += g
        others = {}
        for q, oes in qs.items():
            if q == main:
                continue
            b = sum(max(0, min(o[1], e[0]) - max(o[0], prev_end)) for o in oes)
            if b:
                names = {}
                for o in oes:
                    ov = max(0, min(o[1], e[0]) - max(o[0], prev_end))
                    if ov:
                        names[o[2][:28]] = names.get(o[2][:28], 0) + ov
                top = sorted(names.items(), key=lambda kv: -kv[1])[:2]
                others[q] = f"{b / 1e3:.0f}us [" + ", ".join(f"{n}:{t / 1e3:.0f}" for n, t in top) + "]"
        print(f"  +{(prev_end - t0) / 1e3:8.1f}  gap {g:7.1f} us  -> {e[2][:44]:44s} | {others}")
    prev_end = max(prev_end, e[1])
print(f"main queue: {sum(x[1] - x[0] for x in es) / 1e3:.1f} us of kernels, {tot_gap:.1f} us in gaps >= {gap_us:.0f} us")
