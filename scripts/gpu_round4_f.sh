#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-f}
timeout 600 python -m pytest tests/test_gpu_zz_streams.py -q -m gpu -x 2>&1 | tail -60 > gpurun_out/r04${T}_zz.log
grep -n "Error\|passed\|failed" gpurun_out/r04${T}_zz.log | head -20
B="--no-cpu-baseline --no-side --no-fwd --no-kernel-pass"
for cfg in "BEVBERT_STAGE_PINNED=1 BEVBERT_BENCH_DL_PIN=0" "BEVBERT_STAGE_PINNED=0 BEVBERT_BENCH_DL_PIN=1" "BEVBERT_STAGE_PINNED=1 BEVBERT_BENCH_DL_PIN=1" "BEVBERT_STAGE_PINNED=0 BEVBERT_BENCH_DL_PIN=0"; do
  env $cfg timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['sustained']; print('$cfg', d['ms_per_step'], {k:s[k] for k in ('ms_per_step','vs_resident','loader_ms_per_batch','loader_wait_ms_per_batch','producer_waits_for_collate_ms_per_batch','producer_waits_for_consumer_ms_per_batch')})" | tee -a gpurun_out/r04${T}_loader_ab.txt
done
for ef in 1 0 1 0; do
  BEVBERT_EARLY_FLUSH=$ef timeout 300 python bench.py $B --no-stream 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('EARLY_FLUSH=$ef', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r04${T}_early_flush_ab.txt
done
