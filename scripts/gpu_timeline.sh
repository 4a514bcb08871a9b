#!/bin/bash
# Kernel trace of the default bench run + scripts/trace_timeline.py over one step (GPU box).
set -u
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/timeline_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o bench -- python $ROOT/bench.py --steps 11 --warmup 11 --no-cpu-baseline --no-kernel-pass "$@" > "$OUT/bench.log" 2>&1
for s in 2 3 4; do python3 $ROOT/scripts/trace_timeline.py "$OUT/trace" $s 40 > "$OUT/timeline_step_minus$s.txt" 2>&1; done
python3 $ROOT/scripts/summarize_trace.py "$OUT/trace" 11 > "$OUT/steps_summary.txt" 2>&1
rm -rf "$OUT/trace"
