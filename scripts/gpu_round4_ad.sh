#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x -k "spills_into" 2>&1 | tail -15 | tee gpurun_out/r04ad_tests.log
