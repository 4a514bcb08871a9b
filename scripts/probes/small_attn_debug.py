import os, sys, math, torch
os.environ["BEVBERT_ATTN_SMALL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from vln_bevbert_amd import ops
import test_gpu_kernels as T
for (Lq, Lk, mk, p) in [(70, 49, "neg", 0.0), (70, 49, "neg", 0.1), (70, 49, None, 0.1), (70, 48, "neg", 0.1), (64, 49, "neg", 0.1)]:
    B, dtype = 3, torch.bfloat16
    q, k, v, km, _, nh = T._make_attn_inputs(B, Lq, Lk, mk, False, dtype, seed=Lq + Lk)
    ops.RT.new_step(1234 + Lk)
    qi, ki, vi = (t.clone().requires_grad_(True) for t in (q, k, v))
    o = ops._Attention.apply("sep", qi, ki, vi, km, None, nh, p, 2)
    Lk2 = (Lk + 1) // 2 * 2
    keep = None
    if p > 0:
        keep = ops.dropout_keep_mask(B * nh * Lq * Lk2, p, ops.RT.seed, 0, "cuda").view(B, nh, Lq, Lk2)[..., :Lk]
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    orf = T._attn_ref(qr, kr, vr, km, None, nh, keep, p)
    do = torch.randn_like(orf).to(dtype)
    o.backward(do); orf.backward(do.float())
    print(Lq, Lk, mk, p, "fwd", float((o.float() - orf).abs().max()))
    for name, a, b_ in (("dq", qi.grad, qr.grad), ("dk", ki.grad, kr.grad), ("dv", vi.grad, vr.grad)):
        e = (a.float() - b_).abs()
        L = a.shape[1]
        per_row = e.view(B, L, nh, 64).amax(dim=(0, 2, 3))
        bad = (per_row > 0.05 * float(b_.abs().max())).nonzero().flatten().tolist()
        print("  ", name, "rel", float(T.rel_err(a, b_)), "bad rows", bad[:40])
