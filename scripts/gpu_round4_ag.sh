#!/bin/bash
# round 4: runtime knobs, one call: HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory) against the default.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
B="--no-cpu-baseline --no-side --no-stream --no-fwd --no-kernel-pass"
rm -f gpurun_out/r04ag_env_ab.txt
for rep in 1 2; do
  for v in default devkernarg; do
    if [ $v = devkernarg ]; then export HIP_FORCE_DEV_KERNARG=1; else unset HIP_FORCE_DEV_KERNARG; fi
    timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('env=$v', d['value'], d['ms_per_step'], d.get('launch_calibration'))" | tee -a gpurun_out/r04ag_env_ab.txt
  done
done
