#!/bin/bash
# round 4, end-of-round record (one gpurun call): full GPU suite, smoke, default bench, rocprofv3 kernel trace of the bench
# command, SQ counters and HBM traffic (separate --pmc passes) of the attention kernels, rollout benches.   usage: ... <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-z}
rm -f gpurun_out/bf16_errors.jsonl
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/r04${T}_gputests.log
tail -4 gpurun_out/r04${T}_gputests.log
cp gpurun_out/bf16_errors.jsonl gpurun_out/r04${T}_bf16_errors.jsonl 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 > gpurun_out/r04${T}_smoke.log; tail -2 gpurun_out/r04${T}_smoke.log
timeout 900 python bench.py > gpurun_out/r04${T}_bench.json 2> gpurun_out/r04${T}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04${T}_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("entry_frac"), "cpu", d.get("cpu_baseline", {}).get("value"))
print("sustained", {k: d["sustained"].get(k) for k in ("samples_per_s", "vs_resident", "loader_ms_per_batch")})
print("side", {k: (v.get("value") or v.get("ms_per_nav_step") or v) for k, v in d.get("side_configs", {}).items()})
PY
bash scripts/gpu_profile.sh r04${T} --no-stream --no-side --no-fwd > /dev/null 2>&1
cd "$ROOT"
head -14 gpurun_out/prof_r04${T}/steps_summary.txt
bash scripts/gpu_pmc_attn2.sh r04${T} 64 441 441 0.1 > /dev/null 2>&1
bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
cd "$ROOT"
cat gpurun_out/pmc_traffic/attn_traffic.json | head -20
rm -f gpurun_out/r04${T}_nav.jsonl
for args in "--map device" "--map device --feedback" "--map host" "--map host --feedback" \
            "--map device --mode train --iters 3 --warmup 2" "--map host --mode train --iters 3 --warmup 2"; do
  timeout 300 python scripts/bench_nav.py --steps 15 --iters 6 --warmup 4 $args 2>&1 | tail -1 >> gpurun_out/r04${T}_nav.jsonl
done
cut -c1-420 gpurun_out/r04${T}_nav.jsonl
