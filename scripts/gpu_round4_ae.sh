#!/bin/bash
# round 4: instruction K|V cached per episode in the rollout runner: graphs-vs-eager check, nav tests, timing.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python scripts/bench_nav.py --batch 32 --steps 15 --check 2>&1 | tail -2 | cut -c1-600 | tee gpurun_out/r04ae_check.log
timeout 600 python -m pytest tests -q -m gpu -x -k "rollout or nav" 2>&1 | tail -3 | tee gpurun_out/r04ae_tests.log
rm -f gpurun_out/r04ae_nav.jsonl
for args in "--map device" "--map device --feedback"; do
  timeout 300 python scripts/bench_nav.py --steps 15 --iters 6 --warmup 4 $args 2>&1 | tail -1 >> gpurun_out/r04ae_nav.jsonl
  BEVBERT_NAV_TEXT_CACHE=0 timeout 300 python scripts/bench_nav.py --steps 15 --iters 6 --warmup 4 $args 2>&1 | tail -1 >> gpurun_out/r04ae_nav.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r04ae_nav.jsonl"):
    d = json.loads(l); print(d["action_feedback"][:6], d["ms_per_nav_step"], d["host_map_bookkeeping_ms_per_nav_step"], d.get("text_kv_cache"))
PY
