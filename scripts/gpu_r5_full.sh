#!/bin/bash
# round 5: full GPU check of HEAD -- every -m gpu test, smoke(), one default bench line.  usage: gpu_r5_full.sh <tag> [bench args]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-x}; shift || true
O=gpurun_out/r05${T}
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 > ${O}_gputests.log; tail -3 ${O}_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 > ${O}_smoke.log; tail -1 ${O}_smoke.log
timeout 900 python bench.py "$@" > ${O}_bench.json 2> ${O}_bench.err || tail -5 ${O}_bench.err
python - <<PY
import json
d = json.loads(open("${O}_bench.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "steps")}, r.get("kernel"), r.get("frac"), r.get("entry_frac"), r.get("avg_launch_us"),
      "sustained", (d.get("sustained") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
print("fwd", d.get("fwd_ms_per_batch"))
PY
