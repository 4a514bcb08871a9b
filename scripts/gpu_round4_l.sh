#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -k "word_embedding or embed_sum or tiny_ragged or full_size or device_graph" 2>&1 | tail -6
python - <<'PY'
import torch, time
from vln_bevbert_amd import ops
from vln_bevbert_amd.lib import call, ptr, stream, dtype_code
rows, V, H = 5120, 30522, 768
ids = torch.randint(1000, 29000, (rows,), device="cuda")
d = torch.randn(rows, H, device="cuda").bfloat16()
sink = torch.zeros(V, H, device="cuda")
for _ in range(3): call("bevbert_embedding_grad", ptr(ids), ptr(d), ptr(sink), rows, H, 0, dtype_code(d), stream())
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): call("bevbert_embedding_grad", ptr(ids), ptr(d), ptr(sink), rows, H, 0, dtype_code(d), stream())
e.record(); torch.cuda.synchronize()
print("embedding_grad 5120 rows (uniform ids): %.1f us per launch" % (s.elapsed_time(e) / 20 * 1e3))
PY
B="--no-cpu-baseline --no-side --no-fwd --no-kernel-pass --no-stream"
timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
