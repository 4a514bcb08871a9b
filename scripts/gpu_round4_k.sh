#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python scripts/probes/rerun_diff.py mlm 5 sap 2>&1 | grep -v amdgpu.ids | tail -14 | tee gpurun_out/r04k_rerun_diff.txt
timeout 300 python scripts/probes/rerun_diff.py sap 5 mlm,masksem 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a gpurun_out/r04k_rerun_diff.txt
timeout 300 python scripts/probes/rerun_diff.py masksem 5 sap 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a gpurun_out/r04k_rerun_diff.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -q -m gpu -k "tiny or full_r2r or per_module or nav_api or object_token or embed_sum or full_size" 2>&1 | tail -8
