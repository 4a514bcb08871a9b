#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-d}
timeout 420 python -m pytest tests/test_gpu_kernels.py -q -k "attention" 2>&1 | grep -E "^(FAILED|[0-9]+ (passed|failed))|Error" | cut -c1-200 | head -30 > gpurun_out/r03${T}_tests.log
rm -f gpurun_out/r03${T}_shape.jsonl
for p in 0.1 0.0; do
  timeout 120 python scripts/bench_attn_shape.py 64 441 441 $p 2>/dev/null >> gpurun_out/r03${T}_shape.jsonl
  BEVBERT_FWD2_NW=4 timeout 120 python scripts/bench_attn_shape.py 64 441 441 $p 2>/dev/null >> gpurun_out/r03${T}_shape.jsonl
done
cat gpurun_out/r03${T}_tests.log gpurun_out/r03${T}_shape.jsonl
