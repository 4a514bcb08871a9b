#!/bin/bash
# round 4, call M: the multi-GPU code path on ONE rank (RCCL group of size 1, collectives forced) after this round's changes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
B="--no-cpu-baseline --no-side --no-fwd --no-kernel-pass --no-stream"
for cfg in "BEVBERT_FORCE_COLLECTIVES=1" "BEVBERT_FORCE_COLLECTIVES=1 BEVBERT_GRAD_EXCHANGE=bf16" "BEVBERT_FORCE_COLLECTIVES=1 BEVBERT_GRAD_EXCHANGE=bf16_a2a"; do
  env $cfg timeout 400 python bench.py $B > gpurun_out/r04m_tmp.json 2> gpurun_out/r04m_tmp.err
  tail -1 gpurun_out/r04m_tmp.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('rccl',{}); print('[$cfg]', d['value'], d['ms_per_step'], 'launch:', d.get('step_launch'), 'graph_error:', d.get('graph_error'), 'rccl keys:', list(r.keys())[:8])" | tee -a gpurun_out/r04m_forced_collectives.txt
  cp gpurun_out/r04m_tmp.json "gpurun_out/r04m_bench_$(echo $cfg | tr ' =' '__').json"
done
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 11 --warmup 11 $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[torchrun nproc=1]', d['value'], d['ms_per_step'], d['n_gpus'], d.get('step_launch'))" | tee -a gpurun_out/r04m_forced_collectives.txt
