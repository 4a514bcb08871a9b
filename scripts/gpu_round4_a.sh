#!/bin/bash
# round 4, call A: full GPU suite (no -x: see every failure), smoke, default bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-a}
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r04${T}_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 > gpurun_out/r04${T}_smoke.log
timeout 600 python bench.py > gpurun_out/r04${T}_bench.json 2> gpurun_out/r04${T}_bench.err
tail -15 gpurun_out/r04${T}_gputests.log; tail -2 gpurun_out/r04${T}_smoke.log
python - <<PY
import json
d = json.loads(open("gpurun_out/r04${T}_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d.get("roofline"))
PY
