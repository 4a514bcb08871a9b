#!/bin/bash
# round 3g: independent-waves short backward -- parity, then timings against the single-pass kernel
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -6
: > gpurun_out/r03za_shape.jsonl
for shape in "64 80 80" "320 36 36" "64 17 80" "64 80 17" "64 96 96"; do
  for nb in 1 0; do
    BEVBERT_ATTN_SMALL_BWD=$nb timeout 120 python scripts/bench_attn_shape.py $shape 0.1 50 mask 2>&1 | grep -v amdgpu >> gpurun_out/r03za_shape.jsonl
  done
done
cat gpurun_out/r03za_shape.jsonl | cut -c1-260
