#!/bin/bash
# round 5 diagnostics: ablations of the 7+1-wave backward (BEVBERT_B2_ABL, results are wrong by construction)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-a}
O=gpurun_out/r05${T}_bwd2_ablations.jsonl
: > $O
for A in ${ABLS:-0 1 2 3 4 8 10 16 7 11 15 31}; do
  BEVBERT_B2_ABL=$A timeout 120 python scripts/bench_attn_shape.py 64 441 441 0.1 20 2>&1 | grep '^{' >> $O
done
python - <<PY
import json
for l in open("$O"):
    d = json.loads(l); print(d['env'].get('BEVBERT_B2_ABL'), 'bwd_us', d['bwd_us'])
PY
