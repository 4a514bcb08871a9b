#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: scripts/kernel_resources.py vln_bevbert_amd/csrc/attn_fwd2.hip [name-filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c",
                      "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    if " error: " in line:
        print(line)
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    body = m.group(1)
    if body.startswith("Function Name:"):
        cur = {"name": body.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in body:
        k, v = body.rsplit(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip() or r["name"]
    n = n.replace("(AttnArgs)", "").replace("void ", "")
    if filt and filt not in n:
        continue
    g = lambda k: r.get(k, "?")
    print(f"{n:60s} vgpr {g('VGPRs'):>4s} sgpr {g('TotalSGPRs'):>4s} scratch {g('ScratchSize [bytes/lane]'):>4s} B  "
          f"waves/simd {g('Occupancy [waves/SIMD]')}  lds {g('LDS Size [bytes/block]')}")
