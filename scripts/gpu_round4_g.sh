#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-g}
rm -f gpurun_out/bf16_errors.jsonl
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -60 > gpurun_out/r04${T}_gputests.log
grep -n "Error\|passed\|failed\|FAILED" gpurun_out/r04${T}_gputests.log | head -30
cp gpurun_out/bf16_errors.jsonl gpurun_out/r04${T}_bf16_errors.jsonl 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04${T}_bench.json 2> gpurun_out/r04${T}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04${T}_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, "graph_error", d.get("graph_error"), d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("entry_frac"))
print("sustained", {k: d["sustained"].get(k) for k in ("samples_per_s", "vs_resident", "loader_ms_per_batch", "loader_wait_ms_per_batch")})
print("side", {k: {kk: v.get(kk) for kk in ("value", "ms_per_step", "ms_per_nav_step", "host_map_bookkeeping_ms_per_nav_step", "error")} for k, v in d.get("side_configs", {}).items()})
PY
tail -3 gpurun_out/r04${T}_bench.err
