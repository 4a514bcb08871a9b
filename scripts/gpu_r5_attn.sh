#!/bin/bash
# round 5: attention kernels -- parity tests of the new forward / backward loops, then A/B timings against the round-3/4
# loops in ONE call (BEVBERT_B2_VAR / BEVBERT_FWD_VAR = 0 select the old loops), a phase timeline and one SQ counter pass.
# usage: gpu_r5_attn.sh <tag> [quick]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-a}; MODE=${2:-full}
O=gpurun_out/r05${T}
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -6 > ${O}_attn_tests.log; tail -3 ${O}_attn_tests.log
: > ${O}_attn_ab.jsonl
for P in 0.1 0.0; do
  for V in "1 1" "0 0" "1 0" "0 1"; do
    set -- $V
    BEVBERT_B2_VAR=$1 BEVBERT_FWD_VAR=$2 timeout 120 python scripts/bench_attn_shape.py 64 441 441 $P 30 2>&1 | grep '^{' >> ${O}_attn_ab.jsonl
  done
done
BEVBERT_B2_VAR=1 timeout 120 python scripts/bench_attn_shape.py 64 441 441 0.1 30 mask 2>&1 | grep '^{' >> ${O}_attn_ab.jsonl
for S in "64 80 441" "64 441 80" "64 200 448"; do
  for V in 1 0; do
    BEVBERT_B2_VAR=$V BEVBERT_FWD_VAR=$V timeout 120 python scripts/bench_attn_shape.py $S 0.1 30 2>&1 | grep '^{' >> ${O}_attn_ab.jsonl
  done
done
cat ${O}_attn_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['Lq'], d['Lk'], d['p'], d['mask'], d['env'], 'bits', d.get('bits_us'), 'fwd', d['fwd_us'], d['fwd_frac'], 'bwd', d['bwd_us'], d['bwd_frac'])"
BEVBERT_B2_TRACE=1 timeout 120 python scripts/bench_attn_shape.py 64 441 441 0.1 10 > ${O}_bwd2_timeline.txt 2>&1
head -3 ${O}_bwd2_timeline.txt | cut -c1-300; grep "step  [56]:" ${O}_bwd2_timeline.txt
if [ "$MODE" = full ]; then
  bash scripts/gpu_pmc_attn2.sh r05${T} 64 441 441 0.1 8 > ${O}_pmc.log 2>&1
  cp gpurun_out/pmc_attn_r05${T}/summary.txt ${O}_pmc_attn_sq_counters.txt 2>/dev/null
  grep -A24 "attn_bwd2\|attn_fwd3" ${O}_pmc_attn_sq_counters.txt | grep "kernel\|WAIT\|MFMA_BUSY\|WAVE_CYCLES\|ACTIVE_INST_ANY\|INSTS_VALU\|INSTS_LDS" 
fi
