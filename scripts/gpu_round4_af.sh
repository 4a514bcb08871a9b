#!/bin/bash
# round 4: does a larger hipBLASLt workspace (256 MB instead of 32 MB per stream) admit faster solutions?  Exhaustive
# search with the large workspace -> table; A/B against the shipped table, two runs each, one call.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
B="--no-cpu-baseline --no-side --no-stream --no-fwd --no-kernel-pass"
t0=$(date +%s)
BEVBERT_LT_WS_MB=256 BEVBERT_GEMM_TABLE=/nonexistent BEVBERT_LT_EXHAUSTIVE=1 timeout 600 python bench.py $B --steps 4 --warmup 2 \
  --save-gemm-tuning gpurun_out/r04af_gemm_tuning_ws256.txt > gpurun_out/r04af_tune.json 2> gpurun_out/r04af_tune.err
echo "tuning run: $(( $(date +%s) - t0 )) s, rows: $(wc -l < gpurun_out/r04af_gemm_tuning_ws256.txt)"; tail -2 gpurun_out/r04af_tune.err
rm -f gpurun_out/r04af_ws_ab.txt
for rep in 1 2; do
  for tab in shipped ws256; do
    if [ $tab = ws256 ]; then export BEVBERT_GEMM_TABLE=$ROOT/gpurun_out/r04af_gemm_tuning_ws256.txt BEVBERT_LT_WS_MB=256; else unset BEVBERT_GEMM_TABLE BEVBERT_LT_WS_MB; fi
    timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('table=$tab', d['value'], d['ms_per_step'], d.get('launch_calibration'), 'rejected', d.get('gemm_candidates_rejected_as_not_reproducible'))" | tee -a gpurun_out/r04af_ws_ab.txt
  done
done
diff <(cut -d' ' -f1-12 vln_bevbert_amd/gemm_tuning.txt | head -3) <(cut -d' ' -f1-12 gpurun_out/r04af_gemm_tuning_ws256.txt | head -3) | head -5
