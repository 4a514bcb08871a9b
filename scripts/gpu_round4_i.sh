#!/bin/bash
# round 4, call I: timeline of the captured step (where is the critical path?) + a few stream-layout A/Bs
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
bash scripts/gpu_timeline.sh r04i --no-stream --no-side --no-fwd > /dev/null 2>&1
cd "$ROOT"
head -60 gpurun_out/timeline_r04i/timeline_step_minus2.txt
head -12 gpurun_out/timeline_r04i/steps_summary.txt
B="--no-cpu-baseline --no-side --no-fwd --no-kernel-pass --no-stream"
for cfg in "" "BEVBERT_WGRAD_STREAMS=1" "BEVBERT_WGRAD_STREAMS=3" "BEVBERT_WGRAD_BATCH=4" "BEVBERT_WGRAD_BATCH=10" ""; do
  env $cfg timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r04i_stream_ab.txt
done
