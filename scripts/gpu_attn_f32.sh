#!/bin/bash
# the fp32 MFMA attention (attn_f32.hip): parity tests of the attention entry, then timings against the
# wave-per-row kernels (BEVBERT_ATTN_F32=simple) at the step's shapes.  usage: gpu_attn_f32.sh <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
T=${1:-a}
O=gpurun_out/${T}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -5 > ${O}_f32_attn_tests.log; tail -3 ${O}_f32_attn_tests.log
python - > ${O}_f32_attention_timings.jsonl <<'PY'
import json, math, os, subprocess, sys
code = r'''
import json, math, sys, torch
sys.path.insert(0, ".")
from vln_bevbert_amd import ops
from vln_bevbert_amd.lib import call, dtype_code, ptr, stream
B, nh, H = 64, 12, 768
for Lq, Lk in ((441, 441), (441, 80), (80, 441), (80, 80), (36, 36)):
    for p in (0.0, 0.1):
        torch.manual_seed(0)
        q, k, v, do = (torch.randn(B, L, H, device="cuda") for L in (Lq, Lk, Lk, Lq))
        o, dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        lse = torch.empty(B, nh, Lq, device="cuda"); delta = torch.empty_like(lse)
        st = ops._strides(q, k, v, o)
        fwd = lambda: call("bevbert_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), None, None, st, B, nh, Lq, Lk, 64, 0.125, dtype_code(q), 1, p, 1, 0, None, 0, stream())
        bwd = lambda: call("bevbert_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv), None, None, None, st, B, nh, Lq, Lk, 64, 0.125, dtype_code(q), 1, p, 1, 0, None, stream())
        def t(fn, n):
            fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n): fn()
            e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
        n = 3 if Lq * Lk > 30000 else 10
        fl = 4.0 * B * nh * Lq * Lk * 64
        tf, tb = t(fwd, n), t(bwd, n)
        print(json.dumps({"Lq": Lq, "Lk": Lk, "p": p, "fwd_us": round(tf, 1), "bwd_us": round(tb, 1), "fwd_TFLOPs": round(fl / tf / 1e6, 1), "bwd_TFLOPs": round(2.5 * fl / tb / 1e6, 1)}), flush=True)
'''
for mode in ("mfma", "simple"):
    env = dict(os.environ); env["BEVBERT_ATTN_F32"] = mode
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            d = json.loads(line); d["kernels"] = mode; print(json.dumps(d), flush=True)
    if r.returncode: print(json.dumps({"kernels": mode, "error": r.stderr[-400:]}))
PY
cat ${O}_f32_attention_timings.jsonl | cut -c1-200
