#!/bin/bash
# timing arms only (no tests): gpu_attn_arms.sh <tag> "ENV=.. ENV=.." ...   (B=64, 441x441, p=0.1, 30 iterations each)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-q}; shift || true
O=gpurun_out/${T}_arms.jsonl
: > $O
for ARM in "$@"; do
  env $ARM timeout 120 python scripts/bench_attn_shape.py 64 441 441 ${P:-0.1} 30 2>&1 | grep '^{' >> $O
done
python - <<PY
import json
for l in open("$O"):
    d = json.loads(l); print(d['p'], d['env'], 'fwd', d['fwd_us'], 'bwd', d['bwd_us'], d['bwd_frac'])
PY
