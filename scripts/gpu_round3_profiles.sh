#!/bin/bash
# round 3 profile set (one gpurun call): rocprofv3 kernel trace + stats of the bench command, SQ counters and HBM
# traffic (separate --pmc passes) of the attention kernels at B = 64, 441 x 441, dropout 0.1.   usage: ... <tag>
set -u
T=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
bash scripts/gpu_profile.sh $T --no-stream --no-side --no-fwd > /dev/null 2>&1
bash scripts/gpu_pmc_attn2.sh $T 64 441 441 0.1 > /dev/null 2>&1
bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
cd "$ROOT"
head -30 gpurun_out/prof_$T/summary.txt
cat gpurun_out/pmc_traffic/attn_traffic.json
