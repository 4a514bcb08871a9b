#!/bin/bash
# round 4, call C: device graph map (edge lengths from the host), GEMM reproducibility screening, re-based bf16 gates,
# sustained run with DataLoader workers
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-c}
rm -f gpurun_out/bf16_errors.jsonl
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "device_graph_map" 2>&1 | tail -25 > gpurun_out/r04${T}_tests_gm.log
tail -8 gpurun_out/r04${T}_tests_gm.log
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r04${T}_gputests.log
tail -12 gpurun_out/r04${T}_gputests.log
cp gpurun_out/bf16_errors.jsonl gpurun_out/r04${T}_bf16_errors.jsonl 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --no-side > gpurun_out/r04${T}_bench.json 2> gpurun_out/r04${T}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04${T}_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d.get("roofline"))
print("sustained", d.get("sustained"))
print("gemm", {k: v for k, v in d.items() if "gemm" in k})
PY
tail -5 gpurun_out/r04${T}_bench.err
for args in "--map device --feedback" "--map device"; do
  timeout 300 python scripts/bench_nav.py --steps 15 --iters 6 --warmup 4 $args 2>&1 | tail -1 | cut -c1-900
done
