#!/bin/bash
# round 6: parity tests of the short attention kernels, then their timings against the kernels of rounds 2-5
# usage: gpu_attn_short.sh <tag> ["env assignments" ...]   each extra argument is one more A/B arm
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-s}; shift || true
O=gpurun_out/${T}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -60 > ${O}_attn_tests.log; tail -3 ${O}_attn_tests.log
: > ${O}_short.jsonl
for ARM in "" "BEVBERT_ATTN_SHORT=0" "$@"; do
  for P in 0.1 0.0; do
    env $ARM timeout 300 python scripts/bench_attn_short.py --p $P 2>&1 | grep '^{' >> ${O}_short.jsonl
  done
done
python - <<PY
import json
for l in open("${O}_short.jsonl"):
    d = json.loads(l)
    print(f"{d['B']:4d}x{d['Lq']:3d}x{d['Lk']:3d} p={d['p']} {str(d['env']):40s} fwd {d['fwd_us']:7.2f} us {d['fwd_kernel']:16s} (hbm {d['fwd_hbm_frac']:.2f})  "
          f"bwd {d['bwd_us']:7.2f} us {d['bwd_kernel']:16s} (hbm {d['bwd_hbm_frac']:.2f})")
PY
