#!/bin/bash
# round 6 diagnosis: where the live-loader step loses time against resident batches.  BEVBERT_STEP_EVENTS=1 makes bench.py
# record device events around every step: device time per step and the idle gap in front of the next one.
# arms: priority of the loader's copy stream (own hardware queue at a non-default priority), number of hardware queues.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
python -c "import torch; print('stream priority range', torch.cuda.Stream.priority_range())"
for ARM in "$@"; do
  env $ARM BEVBERT_STEP_EVENTS=1 timeout 600 python bench.py --no-side --no-fwd --no-cpu-baseline --no-kernel-pass --no-sustained-ragged --steps 33 --warmup 22 --detail gpurun_out/ab_detail.json > /dev/null 2> gpurun_out/ab.err
  python - "$ARM" <<P
import json,sys
d=json.loads(open("gpurun_out/ab_detail.json").read())
s=d.get("sustained") or {}
print(sys.argv[1], "| resident", d["ms_per_step"], (d.get("step_events_resident") or {}).get("device_ms_per_step"), "| sustained", s.get("ms_per_step"), s.get("vs_resident"), s.get("step_events"), s.get("arena_fill"), flush=True)
P
done
