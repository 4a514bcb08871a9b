#!/bin/bash
# The GPU record of a code state, one gpurun call.  usage: gpu_record.sh <tag> [part ...]
#   parts (default: all, in this order):
#     tests    every -m gpu test (+ the bf16 error ratios the model tests log)
#     smoke    __graft_entry__.smoke()
#     bench    the default `python bench.py` line (resident + sustained + ragged sustained + side configs + cpu baseline)
#     profile  rocprofv3 --kernel-trace --stats of the bench command, per-step / per-shape tables (summarize_trace.py)
#     sq       SQ counters of the attention kernels at the BEV self-attention shape (separate --pmc passes)
#     traffic  FETCH_SIZE / WRITE_SIZE per launch of every hand-written HBM-bound kernel against its algorithmic bytes
#     modes    the RCCL exchange forced on a one-rank group (bench.py line with the per-region timeline); the fp32 mode and the
#              fp32-residual bf16 mode are side configurations of the default bench line
#     nav      fine-tune rollouts (bench_nav.py): device / host map, with / without action feedback, training mode
#   everything lands in gpurun_out/<tag>_*; copy what is to be kept into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-rec}; shift || true
PARTS=${*:-tests smoke bench profile sq traffic modes nav}
O=gpurun_out/${T}
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line in", sys.argv[1], e); sys.exit(0)
r = d.get("roofline", {})
print({k: d.get(k) for k in ("value", "ms_per_step", "dtype")}, r.get("kernel"), r.get("frac"),
      "bwd", (d.get("roofline_attn_bwd") or {}).get("frac"), "fwd", (d.get("roofline_attn_fwd") or {}).get("frac"),
      "sustained", (d.get("sustained") or {}).get("vs_resident"), "ragged", (d.get("sustained_ragged") or {}).get("vs_resident"),
      "cpu", (d.get("cpu_baseline") or {}).get("value"))
for k, v in (d.get("side_configs") or {}).items():
    print("  side", k, v.get("value") or v.get("ms_per_nav_step") or str(v)[:120])
PY
}
for P in $PARTS; do
  cd "$ROOT"
  case $P in
    tests)
      rm -f gpurun_out/bf16_errors.jsonl
      timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25 > ${O}_gputests.log; tail -3 ${O}_gputests.log
      cp gpurun_out/bf16_errors.jsonl ${O}_bf16_errors.jsonl 2>/dev/null ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 > ${O}_smoke.log; tail -1 ${O}_smoke.log ;;
    bench)
      timeout 1200 python bench.py > ${O}_bench.json 2> ${O}_bench.err || tail -5 ${O}_bench.err
      line ${O}_bench.json ;;
    profile)
      bash scripts/gpu_profile.sh ${T} --no-stream --no-side --no-fwd > /dev/null 2>&1
      cd "$ROOT"; head -12 gpurun_out/prof_${T}/steps_summary.txt ;;
    sq)
      bash scripts/gpu_pmc_attn2.sh ${T} 64 441 441 0.1 > /dev/null 2>&1
      cd "$ROOT"; head -40 gpurun_out/pmc_attn_${T}/summary.txt ;;
    traffic)
      bash scripts/gpu_pmc_all.sh ${T} 2>&1 | tail -40 ;;
    modes)
      BEVBERT_FORCE_COLLECTIVES=1 timeout 600 python bench.py --no-side --no-cpu-baseline --no-stream > ${O}_bench_forced_collectives.json 2> ${O}_bench_forced.err; line ${O}_bench_forced_collectives.json ;;
    nav)
      rm -f ${O}_nav.jsonl
      for args in "--map device" "--map device --feedback" "--map host" "--map host --feedback" \
                  "--map device --mode train --iters 3 --warmup 2" "--map host --mode train --iters 3 --warmup 2"; do
        timeout 300 python scripts/bench_nav.py --steps 15 --iters 6 --warmup 4 $args 2>&1 | tail -1 >> ${O}_nav.jsonl
      done
      cut -c1-300 ${O}_nav.jsonl ;;
    *) echo "unknown part $P" ;;
  esac
done
