#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python scripts/probes/overlap_adamw_probe.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r04n_overlap_adamw_probe.txt
