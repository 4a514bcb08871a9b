#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of a short bench run; with PMC=1 also two counter
# passes (FETCH_SIZE, WRITE_SIZE: separate passes, never combined with other tracing domains) for HBM traffic.
# Usage: [PMC=1] scripts/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 11 --warmup 11 --no-cpu-baseline --no-kernel-pass $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/trace_bench.log" 2>&1
echo "trace rc=$?" >> "$OUT/trace_bench.log"
find "$OUT/trace" -name '*kernel_stats*' -exec cp {} "$OUT/" \; 2>/dev/null
python "$ROOT/scripts/summarize_rocprof.py" "$OUT" > "$OUT/summary.txt" 2>&1
python "$ROOT/scripts/summarize_trace.py" "$OUT/trace" 11 > "$OUT/steps_summary.txt" 2>&1
rm -rf "$OUT/trace"
if [ "${PMC:-0}" = "1" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- \
      python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-pass "$@" > "$OUT/pmc_$C.log" 2>&1
    echo "pmc $C rc=$?" >> "$OUT/pmc_$C.log"
    find "$OUT/pmc_$C" -name '*counter_collection*' -exec cp {} "$OUT/pmc_${C}_counters.csv" \; 2>/dev/null
    rm -rf "$OUT/pmc_$C"
  done
  python "$ROOT/scripts/summarize_pmc.py" "$OUT" > "$OUT/pmc_summary.txt" 2>&1
  rm -f "$OUT"/pmc_*_counters.csv
fi
