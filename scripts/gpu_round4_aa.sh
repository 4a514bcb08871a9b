#!/bin/bash
# round 4: DeviceGraphMap after the host-side diet (digest, pointer arguments, adjacency matrix, gather-views launch):
# its GPU tests, then the rollout bench with and without action feedback.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x -k "graph_map or gather_views or host_feed or rollout or nav" 2>&1 | tail -6 | tee gpurun_out/r04aa_tests.log
for args in "--map device" "--map device --feedback"; do
  timeout 300 python scripts/bench_nav.py --steps 15 --iters 6 --warmup 4 $args 2>&1 | tail -1 >> gpurun_out/r04aa_nav.jsonl
done
cut -c1-600 gpurun_out/r04aa_nav.jsonl
