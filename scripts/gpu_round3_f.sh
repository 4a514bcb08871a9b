#!/bin/bash
# round 3f: short-key attention kernels -- parity tests, then per-shape timings against the tiled kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -15 > gpurun_out/r03y_tests.log
cat gpurun_out/r03y_tests.log
: > gpurun_out/r03y_shape.jsonl
for shape in "64 80 80" "64 441 80" "320 36 36" "64 17 80" "64 80 17"; do
  for small in 1 0; do
    BEVBERT_ATTN_SMALL=$small timeout 120 python scripts/bench_attn_shape.py $shape 0.1 50 mask >> gpurun_out/r03y_shape.jsonl 2>&1
  done
done
for nw in 1 2 3; do
  BEVBERT_SMALL_NW=$nw timeout 120 python scripts/bench_attn_shape.py 64 80 80 0.1 50 mask >> gpurun_out/r03y_shape.jsonl 2>&1
done
for cfg in "7 1" "4 1" "2 2" "4 4"; do
  set -- $cfg
  BEVBERT_SMALL_NW=$1 BEVBERT_SMALL_QPW=$2 timeout 120 python scripts/bench_attn_shape.py 64 441 80 0.1 50 mask >> gpurun_out/r03y_shape.jsonl 2>&1
done
cat gpurun_out/r03y_shape.jsonl
