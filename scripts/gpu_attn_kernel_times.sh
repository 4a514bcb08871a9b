#!/bin/bash
# Kernel-level durations of the attention entry points at every shape of the R2R step (the Python micro-benchmark is
# host-bound on the small shapes): rocprofv3 kernel trace of scripts/bench_attn.py, averaged per (kernel, grid).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/attn_kernels_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o attn -- python $ROOT/scripts/bench_attn.py > "$OUT/bench.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace*.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_" not in n:
            continue
        wg = max(1, int(r.get("Workgroup_Size_X", 1) or 1))
        grid = int(r.get("Grid_Size_X", 0) or 0) // wg
        acc[(n.split("(")[0].replace("void ", ""), grid, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(os.path.join(d, "summary.txt"), "w") as out:
    for (n, grid, wg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        out.write(f"{n:60s} grid {grid:6d} x {wg:4d}  calls {len(v):4d}  median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f} us\n")
print(open(os.path.join(d, "summary.txt")).read())
PY
rm -rf "$OUT/trace"
