#!/bin/bash
# same-box A/B of the two bf16 modes of the bench step: plain bf16 vs bf16 with the fp32 residual stream (torch.autocast's
# arithmetic), alternating, N rounds.  usage: gpu_ab_residual.sh <tag> [rounds=3]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-ab}; N=${2:-3}
: > gpurun_out/${T}_ab_residual.jsonl
for i in $(seq 1 $N); do
  for R in bf16 fp32; do
    timeout 600 python bench.py --residual $R --steps 110 --warmup 22 --no-side --no-stream --no-cpu-baseline --no-kernel-pass --no-fwd 2>/dev/null | tail -1 >> gpurun_out/${T}_ab_residual.jsonl
  done
done
python - <<PY
import json
rows = [json.loads(l) for l in open("gpurun_out/${T}_ab_residual.jsonl") if l.startswith("{")]
for d in rows:
    print(d["dtype"], d["ms_per_step"], d["value"])
by = {}
for d in rows:
    by.setdefault(d["dtype"], []).append(d["ms_per_step"])
for k, v in by.items():
    print(k, "mean", round(sum(v) / len(v), 3), "min", min(v), "max", max(v))
PY
