#!/bin/bash
# quick check of an attention kernel edit -- parity tests of the attention entry, then timings of the big shape
# usage: gpu_attn_quick.sh <tag> ["env assignments" ...]   each extra argument is one A/B arm, e.g. "BEVBERT_ATTN_BWD3=0"
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-q}; shift || true
O=gpurun_out/${T}
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -4 > ${O}_attn_tests.log; tail -2 ${O}_attn_tests.log
: > ${O}_attn.jsonl
for ARM in "" "$@"; do
  for P in 0.1 0.0; do
    env $ARM timeout 120 python scripts/bench_attn_shape.py 64 441 441 $P 30 2>&1 | grep '^{' >> ${O}_attn.jsonl
  done
done
python - <<PY
import json
for l in open("${O}_attn.jsonl"):
    d = json.loads(l); print(d['p'], d['env'], 'bits', d.get('bits_us'), 'fwd', d['fwd_us'], d['fwd_frac'], 'bwd', d['bwd_us'], d['bwd_frac'])
PY
