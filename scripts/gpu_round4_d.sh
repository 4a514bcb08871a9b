#!/bin/bash
# round 4, call D: exhaustive hipBLASLt search (all solutions, not only the heuristic's top 32) -> tuning table; A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-d}
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_zz_streams.py -q -m gpu -k "full_r2r_config or failed_capture or streaming_loader or per_module or tiny_ragged" 2>&1 | tail -12
B="--no-cpu-baseline --no-side --no-stream --no-fwd --no-kernel-pass"
t0=$(date +%s)
BEVBERT_GEMM_TABLE=/nonexistent BEVBERT_LT_EXHAUSTIVE=1 timeout 900 python bench.py $B --steps 4 --warmup 2 \
  --save-gemm-tuning gpurun_out/r04${T}_gemm_tuning_exhaustive.txt > gpurun_out/r04${T}_tune.json 2> gpurun_out/r04${T}_tune.err
echo "tuning run: $(( $(date +%s) - t0 )) s, rows: $(wc -l < gpurun_out/r04${T}_gemm_tuning_exhaustive.txt)"; tail -3 gpurun_out/r04${T}_tune.err
for rep in 1 2; do
  for tab in shipped exhaustive; do
    if [ $tab = exhaustive ]; then export BEVBERT_GEMM_TABLE=$ROOT/gpurun_out/r04${T}_gemm_tuning_exhaustive.txt; else unset BEVBERT_GEMM_TABLE; fi
    timeout 400 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('table=$tab', d['value'], d['ms_per_step'], d.get('launch_calibration'), 'rejected', d.get('gemm_candidates_rejected_as_not_reproducible'))" | tee -a gpurun_out/r04${T}_table_ab.txt
  done
done
export BEVBERT_GEMM_TABLE=$ROOT/gpurun_out/r04${T}_gemm_tuning_exhaustive.txt
timeout 500 python bench.py --no-cpu-baseline --no-side > gpurun_out/r04${T}_bench_exhaustive.json 2> gpurun_out/r04${T}_bench_exhaustive.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04${T}_bench_exhaustive.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, "sustained", {k: d["sustained"][k] for k in ("samples_per_s", "vs_resident", "loader_ms_per_batch", "loader_wait_ms_per_batch")})
k = d["kernels"]
print("gemm ms", k["library_gemm_ms"], "tflops", k["library_gemm_tflops"], "custom", k["custom_kernel_ms"])
for n, r in list(k["by_gemm"].items())[:16]: print(f"  {n:40s} n={r['launches']:3d} avg={r['avg_us']:7.1f}us tf={r['tflops']:.0f}")
PY
