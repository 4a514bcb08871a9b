#!/bin/bash
# round 4: DeviceGraphMap with the BEV arrays riding in nav_gmap_variable's transfer: GPU tests of the map, rollout bench
# with feedback (host bookkeeping figure), and the training-mode rollout (eager, backward through the steps) on both maps.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x -k "graph_map or gather_views or host_feed or rollout or nav" 2>&1 | tail -4 | tee gpurun_out/r04ab_tests.log
rm -f gpurun_out/r04ab_nav.jsonl
for args in "--map device --feedback" "--map device --feedback" "--map device --mode train --iters 3 --warmup 2" "--map host --mode train --iters 3 --warmup 2"; do
  timeout 300 python scripts/bench_nav.py --steps 15 --iters 6 --warmup 4 $args 2>&1 | tail -1 >> gpurun_out/r04ab_nav.jsonl
done
cut -c1-700 gpurun_out/r04ab_nav.jsonl
