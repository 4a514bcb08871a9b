#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_zz_streams.py -q -m gpu -k "tiny or full_r2r or captured_step or static_batch or streaming or reproducible or full_size or rccl" 2>&1 | tail -30 > gpurun_out/r04j_tests.log
grep -n "Error\|passed\|failed\|FAILED" gpurun_out/r04j_tests.log | head -20
B="--no-cpu-baseline --no-side --no-fwd --no-kernel-pass --no-stream"
for cfg in "BEVBERT_EARLY_BEV=1" "BEVBERT_EARLY_BEV=0" "BEVBERT_EARLY_BEV=1" "BEVBERT_EARLY_BEV=0" "BEVBERT_EARLY_BEV=1 BEVBERT_WGRAD_BATCH=10" "BEVBERT_EARLY_BEV=1 BEVBERT_WGRAD_BATCH=8"; do
  env $cfg timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r04j_early_bev_ab.txt
done
