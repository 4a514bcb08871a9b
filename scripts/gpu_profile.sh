#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of a short bench run, then HBM-traffic counters
# in a separate pass (PMC must not be combined with tracing domains other than kernel-trace).
# Usage: scripts/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- \
  python "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-pass "$@" > "$OUT/trace_bench.log" 2>&1
echo "trace rc=$?" >> "$OUT/trace_bench.log"
# keep only the summaries (the raw trace is large)
find "$OUT/trace" -name '*kernel_stats*' -exec cp {} "$OUT/" \; 2>/dev/null
find "$OUT/trace" -name '*kernel_trace*' -size -40M -exec cp {} "$OUT/" \; 2>/dev/null
python "$ROOT/scripts/summarize_rocprof.py" "$OUT" > "$OUT/summary.txt" 2>&1
rm -rf "$OUT/trace"
