#!/bin/bash
# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, SEPARATE passes, kernel trace only) of
#   (a) the attention kernels at the BEV self-attention shape (scripts/bench_attn_shape.py 64 441 441 0.1), and
#   (b) every HBM-bound hand-written kernel at its step size (scripts/bench_rowops.py with marker launches between records),
# joined with the algorithmic bytes of each record -> gpurun_out/r05<tag>_pmc_traffic.json (ratio traffic / algorithmic).
# PMC_PARTS=attn: part (a) only.
# FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide coalesced reads as 64 bytes: MI355X_MICROARCH.md).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-a}
OUT=$ROOT/gpurun_out/pmc_$T
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/attn_$C" -o pmc -- \
    python $ROOT/scripts/bench_attn_shape.py 64 441 441 0.1 6 > "$OUT/attn_$C.log" 2>&1
  find "$OUT/attn_$C" -name '*counter_collection*' -exec cp {} "$OUT/attn_${C}.csv" \;
  rm -rf "$OUT/attn_$C"
  if [ "${PMC_PARTS:-all}" != "attn" ]; then
  ROWOPS_MARK=1 timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/row_$C" -o pmc -- \
    python $ROOT/scripts/bench_rowops.py 3 > "$OUT/row_$C.log" 2>&1
  find "$OUT/row_$C" -name '*counter_collection*' -exec cp {} "$OUT/row_${C}.csv" \;
  rm -rf "$OUT/row_$C"
  fi
done
python3 - "$OUT" "$ROOT/gpurun_out/${T}_pmc_traffic.json" <<'PY'
import csv, sys, os, collections, json
d, dst = sys.argv[1], sys.argv[2]
out = {"_source": "scripts/gpu_pmc_all.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes); bytes per launch = "
                  "(2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950 wide-read correction); algorithmic = the record's byte count"}
# (a) attention: per kernel symbol
att = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, cnt = collections.Counter(), collections.Counter()
    p = os.path.join(d, f"attn_{c}.csv")
    if not os.path.exists(p): continue
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] != c or "attn_" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in tot: att[k][c + "_KiB"] = tot[k] / cnt[k]
B, nh, L, es = 64, 12, 441, 2
alg = {"attn_fwd": (2 * L + 2 * L) * nh * 64 * B * es, "attn_bwd": (4 * L + 4 * L) * nh * 64 * B * es,
       "attn_drop_bits": 3 * 8 * B * nh * 32 * 7 * 16}
for k, v in att.items():
    t = (2 * v.get("FETCH_SIZE_KiB", 0.0) + v.get("WRITE_SIZE_KiB", 0.0)) * 1024
    a = next((x for n, x in alg.items() if n in k), None)
    out["attention:" + k] = {"traffic_bytes": round(t), "algorithmic_bytes": a, "ratio": round(t / a, 3) if a else None, **{x: round(y, 1) for x, y in v.items()}}
# (b) row kernels: counter rows between marker launches (cast_f32_kernel, grid grows with the record index)
recs = {}
for line in (open(os.path.join(d, "row_FETCH_SIZE.log")) if os.path.exists(os.path.join(d, "row_FETCH_SIZE.log")) else []):
    if line.startswith("{"):
        r = json.loads(line); recs[r["mark"]] = r
per = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(lambda: collections.Counter())
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(d, f"row_{c}.csv")
    if not os.path.exists(p): continue
    rows = [r for r in csv.DictReader(open(p)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    grids = sorted({int(r["Grid_Size"]) for r in rows if "cast_f32" in r["Kernel_Name"]})
    cur = 0
    for r in rows:
        if "cast_f32" in r["Kernel_Name"]:
            cur = grids.index(int(r["Grid_Size"])) + 1
            continue
        n = r["Kernel_Name"]
        if cur == 0 or any(x in n for x in ("elementwise", "at::native", "fill", "rand", "sort", "scan", "Memset", "copy")): continue
        per[cur][c] += float(r["Counter_Value"]); nl[cur][c] += 1
        per[cur].setdefault("_names", set()) if False else None
for m, r in sorted(recs.items()):
    v = per.get(m, {})
    launches = 8                                   # 5 warm-up + 3 timed calls per record
    t = (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 / launches
    key = f"{r['kernel']}[rows={r['rows']}" + (f",p={r['p']}" if "p" in r else "") + (f",C={r['C']}" if "C" in r else "") + "]"
    out[key] = {"traffic_bytes": round(t), "algorithmic_bytes": r["bytes"], "ratio": round(t / r["bytes"], 3) if r["bytes"] else None,
                "us_under_profiler": r["us"], "kernel_launches_seen": dict(nl.get(m, {}))}
# the same figures under the keys bench.py's kernel pass uses (C-ABI entry [problem size]), the training configuration (p = 0.1)
alias = {"ln_res32_fwd": "bevbert_layernorm_res32_fwd", "ln_res32_bwd": "bevbert_layernorm_res32_bwd",
         "ln_fwd": "bevbert_bias_dropout_residual_layernorm_fwd", "ln_bwd": "bevbert_layernorm_bwd",
         "gelu_fwd": "bevbert_bias_gelu_fwd", "gelu_bwd": "bevbert_bias_gelu_bwd"}
flat = {"adamw_step": "bevbert_adamw_step", "grad_norm_clip": "bevbert_grad_norm_clip", "bev_splat_mean": "bevbert_bev_splat_mean",
        "multi_accum": "bevbert_multi_accum"}
bench_keys = {}
for m, r in sorted(recs.items()):
    key = f"{r['kernel']}[rows={r['rows']}" + (f",p={r['p']}" if "p" in r else "") + (f",C={r['C']}" if "C" in r else "") + "]"
    t = out.get(key, {}).get("traffic_bytes")
    if t is None:
        continue
    if r["kernel"] in alias and r.get("p", 0.1) == 0.1:
        bench_keys[f"{alias[r['kernel']]}[rows={r['rows']}]"] = t
    if r["kernel"] in flat:
        bench_keys[flat[r["kernel"]]] = t
for k, v in out.items():
    if k.startswith("attention:attn_fwd4"):
        bench_keys["bevbert_attn_fwd[Lq=441,Lk=441]"] = v["traffic_bytes"]
    if k.startswith("attention:attn_bwd3"):
        bench_keys["bevbert_attn_bwd[Lq=441,Lk=441]"] = v["traffic_bytes"]
    if k.startswith("attention:attn_drop_bits"):
        bench_keys["bevbert_attn_drop_bits[Lq=441,Lk=441]"] = v["traffic_bytes"]
out["_bench_keys"] = bench_keys
json.dump(out, open(dst, "w"), indent=1)
for k, v in out.items():
    if k != "_source": print(k, v.get("ratio"), v.get("traffic_bytes"), v.get("algorithmic_bytes"))
PY
rm -f "$OUT"/*.csv
