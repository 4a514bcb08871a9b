#!/bin/bash
# end-of-round record (one gpurun call): GPU suite, smoke, default bench, row-kernel microbench, rollout bench, kernel trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-final}
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r03${T}_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 > gpurun_out/r03${T}_smoke.log
timeout 600 python bench.py > gpurun_out/r03${T}_bench.json 2> gpurun_out/r03${T}_bench.err
timeout 200 python scripts/bench_rowops.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03${T}_rowops.jsonl
timeout 200 python scripts/bench_nav.py --steps 15 --iters 6 --warmup 4 2>&1 | tail -1 > gpurun_out/r03${T}_nav.jsonl
timeout 200 python scripts/bench_nav.py --steps 15 --iters 6 --warmup 4 --no-graphs 2>&1 | tail -1 >> gpurun_out/r03${T}_nav.jsonl
bash scripts/gpu_profile.sh r03${T} --no-stream --no-side --no-fwd > /dev/null 2>&1
cd "$ROOT"
tail -4 gpurun_out/r03${T}_gputests.log; tail -2 gpurun_out/r03${T}_smoke.log
python - <<PY
import json
d = json.loads(open("gpurun_out/r03${T}_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d.get("roofline"))
print("sustained", d.get("sustained", {}).get("samples_per_s"), "side", {k: (v.get("value") or v.get("ms_per_nav_step")) for k, v in d.get("side_configs", {}).items()})
PY
cat gpurun_out/r03${T}_nav.jsonl | cut -c1-400
head -12 gpurun_out/prof_r03${T}/summary.txt
head -3 gpurun_out/prof_r03${T}/steps_summary.txt
