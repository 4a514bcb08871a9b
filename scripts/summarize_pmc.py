#!/usr/bin/env python3
"""Per-kernel average of FETCH_SIZE / WRITE_SIZE from rocprofv3 --pmc passes (counter_collection CSV).
rocprofv3 reports these derived counters in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section), so the table lists both the raw and the x2-corrected fetch figure."""
import csv
import os
import sys
from collections import defaultdict

d = sys.argv[1]
acc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = os.path.join(d, f"pmc_{c}_counters.csv")
    if not os.path.exists(f):
        print("missing", f)
        continue
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") != c:
                continue
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    acc[c] = (tot, cnt)
names = set()
for c in acc:
    names |= set(acc[c][0])
rows = []
for n in names:
    if not any(k in n for k in ("attn_", "ln_fwd", "ln_bwd", "bev_", "colwise", "bias_gelu", "gather_wsum", "adamw",
                                "sumsq", "embedding_grad", "colsum")):
        continue
    f_tot, f_cnt = acc.get("FETCH_SIZE", ({}, {}))
    w_tot, w_cnt = acc.get("WRITE_SIZE", ({}, {}))
    fk = f_tot.get(n, 0.0) / max(1, f_cnt.get(n, 0))
    wk = w_tot.get(n, 0.0) / max(1, w_cnt.get(n, 0))
    rows.append((fk + wk, n, f_cnt.get(n, 0), fk, wk))
print(f"{'launches':>8s} {'fetch_MiB':>10s} {'fetch_x2_MiB':>12s} {'write_MiB':>10s}  kernel (per-launch averages)")
for _, n, k, fk, wk in sorted(rows, reverse=True):
    print(f"{k:8d} {fk / 1024:10.2f} {2 * fk / 1024:12.2f} {wk / 1024:10.2f}  {n[:100]}")
