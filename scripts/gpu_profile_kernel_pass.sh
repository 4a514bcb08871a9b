#!/bin/bash
# rocprofv3 kernel trace of the bench's KERNEL PASS: its last three steps (one per task) are issued eagerly on ONE stream with
# HIP events around every C-ABI launch -- the launches `roofline.avg_launch_us` is averaged over.  The per-step table of the
# replayed steps (gpu_profile.sh) shows the same kernels beside the weight-gradient GEMMs, i.e. slower.
# usage: gpu_profile_kernel_pass.sh <tag>
set -u
TAG=${1:-kp}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_kernel_pass
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o bench -- python $ROOT/bench.py --steps 11 --warmup 11 \
  --no-cpu-baseline --no-stream --no-side --no-fwd --detail $OUT/bench_detail.json > "$OUT/bench_line.json" 2> "$OUT/log.txt"
python "$ROOT/scripts/summarize_trace.py" "$OUT/trace" 3 > "$OUT/kernel_pass_steps.txt" 2>&1
rm -rf "$OUT/trace"
python - "$OUT" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_line.json").read().strip().splitlines()[-1])
print("bench roofline:", d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], "us", d["roofline"]["frac"])
PY
grep -n "ln_res32_bwd\|adamw" "$OUT/kernel_pass_steps.txt" | head -8
