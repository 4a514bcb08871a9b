#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
bash scripts/gpu_profile_nav.sh r04q_nav --feedback 2>&1 | tail -22 | tee gpurun_out/r04q_nav_rollout_steady_state.txt
