#!/bin/bash
# SQ-level counters for the attention kernels at the BEV self-attention shape (B=64, 441x441, dropout 0.1).
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_attn_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/p$i" -o pmc -- \
    python $ROOT/scripts/bench_attn.py one > "$OUT/p$i.log" 2>&1
  echo "rc=$?" >> "$OUT/p$i.log"
  find "$OUT/p$i" -name '*counter_collection*' -exec cp {} "$OUT/p${i}_counters.csv" \;
  rm -rf "$OUT/p$i"
done
python3 - "$OUT" <<'PY'
import csv, sys, os, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in ("p1_counters.csv", "p2_counters.csv"):
    p = os.path.join(d, f)
    if not os.path.exists(p): continue
    seen = collections.Counter()
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if "attn_mfma" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES", "SQ_ACTIVE_INST_LDS"): cnt[(k, f)] += 1
with open(os.path.join(d, "summary.txt"), "w") as out:
    for k, v in acc.items():
        n1 = max(1, cnt[(k, "p1_counters.csv")]); n2 = max(1, cnt[(k, "p2_counters.csv")])
        out.write(f"{k}  launches p1={n1} p2={n2}\n")
        for c, x in sorted(v.items()):
            n = n1 if c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES".split() else n2
            out.write(f"    {c:28s} {x / n:16.0f} per launch\n")
print(open(os.path.join(d, "summary.txt")).read())
PY
rm -f "$OUT"/p*_counters.csv
