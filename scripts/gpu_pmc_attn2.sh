#!/bin/bash
# SQ-level counters of the attention kernels at ONE shape (scripts/bench_attn_shape.py arguments follow the tag).
set -u
TAG=${1:-r03}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_attn_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA"
P3="SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/p$i" -o pmc -- \
    python $ROOT/scripts/bench_attn_shape.py "$@" > "$OUT/p$i.log" 2>&1
  echo "rc=$?" >> "$OUT/p$i.log"
  find "$OUT/p$i" -name '*counter_collection*' -exec cp {} "$OUT/p${i}_counters.csv" \;
  rm -rf "$OUT/p$i"
done
python3 - "$OUT" <<'PY'
import csv, sys, os, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in ("p1_counters.csv", "p2_counters.csv", "p3_counters.csv"):
    p = os.path.join(d, f)
    if not os.path.exists(p): continue
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][-70:]
        if "attn_" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
with open(os.path.join(d, "summary.txt"), "w") as out:
    for k, v in acc.items():
        out.write(f"{k}\n")
        for c, x in sorted(v.items()):
            n = max(1, cnt[k][c])
            out.write(f"    {c:28s} {x / n:16.0f} per launch ({n} launches)\n")
print(open(os.path.join(d, "summary.txt")).read())
PY
rm -f "$OUT"/p*_counters.csv
