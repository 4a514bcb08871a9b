#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-h}
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -k "prenorm or layernorm or per_module or tiny or full_r2r or object_token or nav_api or fp16_autocast or 100_step or training_curve" 2>&1 | tail -40 > gpurun_out/r04${T}_tests.log
grep -n "Error\|passed\|failed\|FAILED" gpurun_out/r04${T}_tests.log | head -20
B="--no-cpu-baseline --no-side --no-fwd --no-kernel-pass --no-stream"
for i in 1 2; do
timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d.get('launch_calibration'))"
done
