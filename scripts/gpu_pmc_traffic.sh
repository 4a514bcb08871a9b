#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the attention kernels at the BEV self-attention shape.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_traffic
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -o pmc -- \
    python $ROOT/scripts/bench_attn_shape.py 64 441 441 0.1 > "$OUT/$C.log" 2>&1
  find "$OUT/$C" -name '*counter_collection*' -exec cp {} "$OUT/${C}.csv" \;
  rm -rf "$OUT/$C"
done
python3 - "$OUT" <<'PY'
import csv, sys, os, collections, json
d = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(os.path.join(d, c + ".csv"))):
        if r["Counter_Name"] != c or "attn_" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in tot: res[k][c + "_KiB_per_launch"] = tot[k] / cnt[k]
for k, v in res.items():
    f = v.get("FETCH_SIZE_KiB_per_launch", 0.0); w = v.get("WRITE_SIZE_KiB_per_launch", 0.0)
    # gfx950: FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM) -> x2
    v["hbm_bytes_per_launch_corrected"] = (2 * f + w) * 1024
json.dump(res, open(os.path.join(d, "attn_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -f "$OUT"/*.csv
