#!/bin/bash
# round 4: backward through a whole rollout (scratch ring spilling into further buffers): its GPU test, then the
# training-mode rollout bench on both maps.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x -k "spills_into or full_size_batch_properties or gather_views" 2>&1 | tail -15 | tee gpurun_out/r04ac_tests.log
rm -f gpurun_out/r04ac_nav.jsonl
for args in "--map device --mode train --iters 3 --warmup 2" "--map host --mode train --iters 3 --warmup 2"; do
  timeout 300 python scripts/bench_nav.py --steps 15 $args > gpurun_out/r04ac_nav.out 2> gpurun_out/r04ac_nav.err
  tail -1 gpurun_out/r04ac_nav.out >> gpurun_out/r04ac_nav.jsonl; tail -3 gpurun_out/r04ac_nav.err
done
cut -c1-700 gpurun_out/r04ac_nav.jsonl
