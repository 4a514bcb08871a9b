#!/bin/bash
# round 4, call E: thread-local capture mode + sticky-error reset, loader with pinned staging, re-based gates
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-e}
timeout 900 python -m pytest tests/test_gpu_zz_streams.py tests/test_gpu_model.py -q -m gpu -k "zz_streams or full_r2r_config or tiny_ragged or object_token" 2>&1 | tail -15
timeout 600 python bench.py --no-cpu-baseline --no-side > gpurun_out/r04${T}_bench.json 2> gpurun_out/r04${T}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04${T}_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, "graph_error", d.get("graph_error"))
print("sustained", d.get("sustained"))
PY
tail -4 gpurun_out/r04${T}_bench.err
