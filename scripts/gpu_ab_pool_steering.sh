#!/bin/bash
# round 6 diagnosis: the collective stream's hardware queue (train.GradReducer.settle_collective_queue) and the eager step's
# side streams (hwqueues.side_stream), on a one-rank group with the exchange forced on, eager steps in the timed region.
# usage: gpu_ab_pool_steering.sh "ENV=.. ENV=.." ...
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
for ARM in "$@"; do
  env $ARM BEVBERT_FORCE_COLLECTIVES=1 timeout 600 python bench.py --launch eager --no-stream --no-side --no-cpu-baseline --no-fwd --no-kernel-pass --steps 33 --warmup 11 --detail gpurun_out/ab_detail.json > /dev/null 2> gpurun_out/ab.err || tail -5 gpurun_out/ab.err
  python - "$ARM" <<P
import json,sys
d=json.loads(open("gpurun_out/ab_detail.json").read())
r=d.get("rccl") or {}
print(sys.argv[1], "eager", d["ms_per_step"], {k: r.get(k) for k in ("collectives_wait_behind_compute","collective_queue_report","eager_ms_per_step_by_exchange","first_collective_at_fraction_of_backward")}, flush=True)
P
done
