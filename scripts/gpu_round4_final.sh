#!/bin/bash
# round 4, last check of HEAD: full GPU suite, smoke, default bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r04final_gputests.log; tail -3 gpurun_out/r04final_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/r04final_smoke.log; tail -1 gpurun_out/r04final_smoke.log
timeout 600 python bench.py --no-side > gpurun_out/r04final_bench.json 2> gpurun_out/r04final_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04final_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d["roofline"]["frac"], d["roofline"].get("entry_frac"), d["roofline"].get("traffic_source", "")[:40], "cpu", d.get("cpu_baseline", {}).get("value"))
PY
