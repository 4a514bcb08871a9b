#!/bin/bash
# round 4, call B: hoisted K/V + device graph map + determinism probe
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-b}
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "device_graph_map or tiny or full_r2r or per_module or nav_api or finetune" 2>&1 | tail -40 > gpurun_out/r04${T}_tests_sel.log
tail -12 gpurun_out/r04${T}_tests_sel.log
for cfg in "" "BEVBERT_WGRAD_STREAMS=1" "BEVBERT_WGRAD_STREAM=0" "BEVBERT_HOIST_KV=0"; do
  echo "== rerun_diff mlm [$cfg]" >> gpurun_out/r04${T}_rerun_diff.txt
  env $cfg timeout 300 python scripts/probes/rerun_diff.py mlm 4 2>&1 | grep -v amdgpu.ids | tail -30 >> gpurun_out/r04${T}_rerun_diff.txt
done
tail -40 gpurun_out/r04${T}_rerun_diff.txt
for args in "--map device" "--map host" "--map device --feedback" "--map host --feedback" "--map device --no-graphs"; do
  timeout 300 python scripts/bench_nav.py --steps 15 --iters 6 --warmup 4 $args 2>&1 | tail -1 >> gpurun_out/r04${T}_nav.jsonl
done
cut -c1-700 gpurun_out/r04${T}_nav.jsonl
timeout 300 python scripts/bench_nav.py --steps 15 --check 2>&1 | tail -2
for h in 1 0 1 0; do
  BEVBERT_HOIST_KV=$h timeout 400 python bench.py --no-cpu-baseline --no-stream --no-side --no-fwd --no-kernel-pass 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('HOIST_KV=$h', d['value'], d['ms_per_step'], d.get('launch_calibration'))" | tee -a gpurun_out/r04${T}_hoist_ab.txt
done
