#!/bin/bash
# rocprofv3 kernel trace + stats of the fine-tune rollout bench (scripts/bench_nav.py), twice (2 and 6 timed episode
# batches after 2 warm-up ones): the DIFFERENCE of the total kernel times is the steady-state GPU time of 4 episode
# batches = 60 navigation steps (first-use GEMM timing happens in the warm-up).  Against the wall time per step it tells
# whether the rollout is bound by the GPU or by the host (map bookkeeping + launches).
set -u
TAG=${1:-nav}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for IT in 2 6; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace$IT" -o nav -- \
    python $ROOT/scripts/bench_nav.py --steps 15 --iters $IT --warmup 2 "$@" > "$OUT/trace_nav_$IT.log" 2>&1
  echo "trace rc=$?" >> "$OUT/trace_nav_$IT.log"
  mkdir -p "$OUT/it$IT"
  find "$OUT/trace$IT" -name '*kernel_stats*' -exec cp {} "$OUT/it$IT/" \; 2>/dev/null
  python "$ROOT/scripts/summarize_rocprof.py" "$OUT/it$IT" > "$OUT/summary_it$IT.txt" 2>&1
  rm -rf "$OUT/trace$IT"
  grep -h "ms_per_nav_step" "$OUT/trace_nav_$IT.log"
  head -5 "$OUT/summary_it$IT.txt"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = {}
for it in (2, 6):
    d = collections.Counter()
    for f in glob.glob(f"{out}/it{it}/*kernel_stats*.csv"):
        for r in csv.DictReader(open(f)):
            d[r["Name"]] += float(r["TotalDurationNs"])
    tot[it] = d
diff = {k: (tot[6][k] - tot[2].get(k, 0.0)) / 60 / 1e3 for k in tot[6]}
print("steady-state kernel time per navigation step: %.2f ms" % (sum(diff.values()) / 1e3))
for k, v in sorted(diff.items(), key=lambda kv: -kv[1])[:14]:
    print("  %8.1f us  %s" % (v, k[:110]))
PY
