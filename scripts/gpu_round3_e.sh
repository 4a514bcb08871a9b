#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-e}
timeout 420 python -m pytest tests/test_gpu_kernels.py -q -k "attention" 2>&1 | grep -E "^(FAILED|[0-9]+ (passed|failed))|Error" | cut -c1-200 | head -30 > gpurun_out/r03${T}_tests.log
rm -f gpurun_out/r03${T}_shape.jsonl
for p in 0.1 0.0; do
  timeout 120 python scripts/bench_attn_shape.py 64 441 441 $p 2>/dev/null >> gpurun_out/r03${T}_shape.jsonl
done
BEVBERT_B2_TRACE=1 timeout 120 python scripts/bench_attn_shape.py 64 441 441 0.1 3 2>/dev/null > gpurun_out/r03${T}_bwd2_trace.txt
cat gpurun_out/r03${T}_tests.log gpurun_out/r03${T}_shape.jsonl
grep -E "wave [0367] step ( [2-4]|13):" gpurun_out/r03${T}_bwd2_trace.txt
