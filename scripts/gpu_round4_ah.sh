#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
timeout 500 python -m pytest tests -q -m gpu -x -k "training_rollout_backward" 2>&1 | tail -12 | tee gpurun_out/r04ah_tests.log
