#!/bin/bash
# full GPU check: every -m gpu test, smoke(), one default bench line.  usage: gpu_full.sh <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-x}
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r03${T}_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 > gpurun_out/r03${T}_smoke.log
timeout 600 python bench.py > gpurun_out/r03${T}_bench.json 2> gpurun_out/r03${T}_bench.err
tail -4 gpurun_out/r03${T}_gputests.log; tail -2 gpurun_out/r03${T}_smoke.log
python - <<PY
import json
d = json.loads(open("gpurun_out/r03${T}_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d.get("roofline"))
PY
