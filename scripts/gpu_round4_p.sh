#!/bin/bash
# round 4, call P: the suite and the default bench once more on another box, at the last commit
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
rm -f gpurun_out/bf16_errors.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r04p_gputests.log
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r04p_gputests.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r04p_bench.json 2> gpurun_out/r04p_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04p_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d["roofline"]["frac"], d["roofline"]["entry_frac"], d["roofline"]["traffic_source"][:40], "cpu", d.get("cpu_baseline", {}).get("value"))
print("sustained", {k: d["sustained"].get(k) for k in ("samples_per_s", "vs_resident", "loader_ms_per_batch", "error")})
print("side", {k: (v.get("value") or v.get("ms_per_nav_step") or v) for k, v in d.get("side_configs", {}).items()})
PY
grep -n "bench +" gpurun_out/r04p_bench.err | tail -3
