"""Per-kernel parity on the MI355X: every C-ABI entry point against the CPU oracle / an fp32 torch reference.

Tolerances (north_star): 1e-3 absolute for fp32 paths (observed ~1e-6), bf16 paths are judged relative to the
output's absolute maximum (1e-2), integer / index work is bit exact.
"""
import math

import numpy as np
import pytest
import torch

from oracle import bevbert_ref as R
from tests.helpers import load_golden
from vln_bevbert_amd import synthetic
from vln_bevbert_amd.config import BevBertConfig

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vln_bevbert_amd import lib, ops as _ops
    lib.load()          # raises (does not skip) when the HIP library is missing on a GPU box
    return _ops


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


# ----------------------------------------------------------------------------- K1 splat
@pytest.mark.parametrize("B,ragged", [(2, True), (5, False)])
def test_lift_bin_cells_bit_exact(ops, B, ragged):
    cfg = BevBertConfig()
    b = synthetic.make_batch(cfg, "sap", B, seed=11 + B, ragged=ragged)
    pc, nod = R.lift_points(b["depths"], b["T_c2w"], b["T_w2c"], b["S_w2c"])
    want = R.cell_index(pc, nod)
    g = synthetic.batch_to(b, DEV)
    pix = ops.pixel_scale(14, DEV)
    cell, order, start = ops.bev_lift_bin(g["depths"], g["T_c2w"], g["T_w2c"], g["S_w2c"], pix, 21, 0.5)
    assert torch.equal(cell.cpu().long(), want), "cell ids differ from the oracle"
    # the order array is a stable sort by cell
    cell_c, order_c, start_c = cell.cpu().numpy(), order.cpu().numpy(), start.cpu().numpy()
    for i in range(B):
        kept = np.flatnonzero(cell_c[i] >= 0)
        exp = kept[np.argsort(cell_c[i][kept], kind="stable")]
        assert start_c[i, -1] == len(kept)
        assert np.array_equal(order_c[i, :len(kept)], exp)
        counts = np.bincount(cell_c[i][kept], minlength=441)
        assert np.array_equal(np.diff(start_c[i]), counts)


def test_splat_mean_bit_exact_and_semantics(ops):
    cfg = BevBertConfig()
    B = 3
    b = synthetic.make_batch(cfg, "sap", B, seed=21, ragged=True)
    want = R.lift_splat(cfg, b)
    g = synthetic.batch_to(b, DEV)
    pix = ops.pixel_scale(14, DEV)
    _, order, start = ops.bev_lift_bin(g["depths"], g["T_c2w"], g["T_w2c"], g["S_w2c"], pix, 21, 0.5)
    feat = g["rgbs"].reshape(B, -1, 768)
    out, sem, semm = ops.bev_splat_mean(feat, order, start, 441, sems=g["sems"].reshape(B, -1, 40))
    assert torch.equal(out.cpu(), want["bev_fts"]), "scatter-mean is not bit-identical to the oracle"
    assert torch.equal(sem.cpu(), want["bev_sems"].to(torch.uint8))
    assert torch.equal(semm.cpu().bool(), want["bev_sem_masks"])
    # compact class ids give the same pooled semantics
    ids = synthetic.make_batch(cfg, "sap", B, seed=21, ragged=True, sems_as="ids")["sems"].to(DEV)
    _, sem2, semm2 = ops.bev_splat_mean(feat, order, start, 441, sems=ids)
    assert torch.equal(sem2, sem) and torch.equal(semm2, semm)
    # fp16 / bf16 feature stores (the on-disk format is fp16): mean of the rounded inputs, fp32 accumulate
    for dt in (torch.float16, torch.bfloat16):
        o, _, _ = ops.bev_splat_mean(feat.to(dt), order, start, 441, out_dtype=torch.float32)
        ref = R.lift_splat(cfg, dict(b, rgbs=b["rgbs"].to(dt).float()))["bev_fts"]
        assert torch.equal(o.cpu(), ref)


@pytest.mark.parametrize("dim,res", [(11, 1.0), (14, 0.5), (21, 0.25)])
def test_lift_bin_other_geometries(ops, dim, res):
    cfg = BevBertConfig(bev_dim=dim, bev_res=res)
    b = synthetic.make_batch(cfg, "sap", 3, seed=dim, ragged=True)
    pc, nod = R.lift_points(b["depths"], b["T_c2w"], b["T_w2c"], b["S_w2c"])
    want = R.cell_index(pc, nod, dim, res)
    g = synthetic.batch_to(b, DEV)
    cell, order, start = ops.bev_lift_bin(g["depths"], g["T_c2w"], g["T_w2c"], g["S_w2c"], ops.pixel_scale(14, DEV),
                                          dim, res)
    assert torch.equal(cell.cpu().long(), want)
    out, _, _ = ops.bev_splat_mean(g["rgbs"].reshape(3, -1, 768), order, start, dim * dim)
    assert torch.equal(out.cpu(), R.lift_splat(cfg, b)["bev_fts"])


def test_splat_golden_edge_cases(ops):
    gld = load_golden("splat_edge")
    pts = torch.from_numpy(gld["pts"])[None].to(DEV)
    nod = torch.from_numpy(gld["no_depth"]).to(DEV)
    feat = torch.from_numpy(gld["feat"]).to(DEV)
    cell, order, start = ops.bev_bin_points(pts, nod, 21, 0.5)
    out, sem, semm = ops.bev_splat_mean(feat, order, start, 441, sems=torch.from_numpy(gld["sem_ids"]).to(DEV).to(torch.uint8))
    assert np.array_equal(out.cpu().numpy(), gld["bev"])
    assert np.array_equal(sem.cpu().numpy(), gld["bev_sems"])
    assert np.array_equal(semm.cpu().numpy().astype(bool), gld["bev_sem_masks"])
    c = cell.cpu()[0]
    assert c[0] == -1 and c[1] == 220 and c[7] == 230 and c[8] == -1 and c[9] == 210 and c[10] == -1


def test_splat_full_size_properties(ops):
    """BASELINE.json full size (B=64): size-independent checks -- every kept point lands in exactly one cell list,
    cell means reproduce the feature sum, empty cells are exactly zero."""
    cfg = BevBertConfig()
    B = 64
    b = synthetic.batch_to(synthetic.make_batch(cfg, "sap", B, seed=77, sems_as="ids"), DEV)
    pix = ops.pixel_scale(14, DEV)
    cell, order, start = ops.bev_lift_bin(b["depths"], b["T_c2w"], b["T_w2c"], b["S_w2c"], pix, 21, 0.5)
    feat = b["rgbs"].reshape(B, -1, 768)
    out, sem, semm = ops.bev_splat_mean(feat, order, start, 441, sems=b["sems"])
    counts = (start[:, 1:] - start[:, :-1]).long()
    kept = (cell >= 0)
    assert torch.equal(counts.sum(1), kept.sum(1))
    # sum_c count_c * mean_c == sum of kept features (fp64 check of a checksum)
    lhs = (out.double() * counts[..., None]).sum((1, 2))
    rhs = (feat.double() * kept[..., None]).sum((1, 2))
    assert float(((lhs - rhs).abs() / rhs.abs().clamp_min(1)).max()) < 1e-4
    assert float(out[counts == 0].abs().max()) == 0.0
    assert torch.equal(semm.bool(), counts > 0)
    occ = float((counts > 0).float().mean())
    assert 0.5 < occ <= 1.0


# ----------------------------------------------------------------------------- K3 / K4 / K5 row kernels
def _ln_ref(x, bias, res, g, b, eps, keep=None, p=0.0):
    z = x.float()
    if bias is not None:
        z = z + bias
    if keep is not None:
        z = torch.where(keep, z / (1 - p), torch.zeros_like(z))
    if res is not None:
        z = z + res.float()
    return torch.nn.functional.layer_norm(z, (z.shape[-1],), g, b, eps), z


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_res,p", [(True, 0.0), (True, 0.1), (False, 0.0), (False, 0.3)])
def test_layernorm_fwd_bwd(ops, dtype, with_res, p):
    torch.manual_seed(0)
    rows, H = 517, 768
    x = torch.randn(rows, H, device=DEV).to(dtype)
    res = torch.randn(rows, H, device=DEV).to(dtype) if with_res else None
    bias = (0.1 * torch.randn(H, device=DEV)).requires_grad_(True)
    gam = (1 + 0.1 * torch.randn(H, device=DEV)).requires_grad_(True)
    bet = (0.1 * torch.randn(H, device=DEV)).requires_grad_(True)
    xr = x.clone().float().requires_grad_(True)
    rr = res.clone().float().requires_grad_(True) if with_res else None
    ops.RT.new_step(1234)
    xin = x.clone().requires_grad_(True)
    rin = res.clone().requires_grad_(True) if with_res else None
    y = ops.bias_dropout_residual_layernorm(xin, bias, rin, gam, bet, 1e-12, p, training=True, inplace_z=False)
    keep = ops.dropout_keep_mask(rows * H, p, ops.RT.seed, 0, DEV).view(rows, H) if p > 0 else None
    if keep is not None:
        assert abs(float(keep.float().mean()) - (1 - p)) < 0.01
    bias_r, gam_r, bet_r = (t.detach().clone().requires_grad_(True) for t in (bias, gam, bet))
    yr, _ = _ln_ref(xr, bias_r, rr, gam_r, bet_r, 1e-12, keep, p)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert float((y.float() - yr).detach().abs().max()) < tol
    dy = torch.randn(rows, H, device=DEV)
    y.backward(dy.to(dtype))
    yr.backward(dy.to(dtype).float())
    gt = 1e-3 if dtype == torch.float32 else 3e-2
    assert rel_err(xin.grad, xr.grad) < gt
    if with_res:
        assert rel_err(rin.grad, rr.grad) < gt
    assert rel_err(gam.grad, gam_r.grad) < gt
    assert rel_err(bet.grad, bet_r.grad) < gt
    assert rel_err(bias.grad, bias_r.grad) < gt


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_prenorm_residual_layernorm_returns_the_stream_and_folds_its_gradient(ops, dtype, p):
    """ops.bias_dropout_residual_prenorm: (LayerNorm(z), z) with z = residual + dropout(x + bias) -- the residual add of a
    pre-norm block and the LayerNorm opening the next sub-layer in one launch (transformer.py:170-182); BOTH outputs are
    used downstream, and the gradient arriving at z is added to LayerNorm's input gradient inside the backward kernel
    (bevbert_layernorm_bwd_add).  Against the torch composition with the exported keep mask; the no-grad call gives the
    same two tensors."""
    torch.manual_seed(1)
    rows, H = 389, 768
    x = torch.randn(rows, H, device=DEV).to(dtype)
    res = torch.randn(rows, H, device=DEV).to(dtype)
    bias = (0.1 * torch.randn(H, device=DEV)).requires_grad_(True)
    gam = (1 + 0.1 * torch.randn(H, device=DEV)).requires_grad_(True)
    bet = (0.1 * torch.randn(H, device=DEV)).requires_grad_(True)
    ops.RT.new_step(77)
    xin, rin = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    y, z = ops.bias_dropout_residual_prenorm(xin, bias, rin, gam, bet, 1e-5, p, training=True)
    keep = ops.dropout_keep_mask(rows * H, p, ops.RT.seed, 0, DEV).view(rows, H) if p > 0 else None
    xr, rr = x.clone().float().requires_grad_(True), res.clone().float().requires_grad_(True)
    bias_r, gam_r, bet_r = (t.detach().clone().requires_grad_(True) for t in (bias, gam, bet))
    yr, zr = _ln_ref(xr, bias_r, rr, gam_r, bet_r, 1e-5, keep, p)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert float((y.float() - yr).detach().abs().max()) < tol and float((z.float() - zr).detach().abs().max()) < tol
    dy, dz = torch.randn(rows, H, device=DEV), torch.randn(rows, H, device=DEV)
    (y.float() * dy + z.float() * dz).sum().backward()
    (yr * dy + zr * dz).sum().backward()
    gt = 1e-3 if dtype == torch.float32 else 3e-2
    for got, want in ((xin.grad, xr.grad), (rin.grad, rr.grad), (gam.grad, gam_r.grad), (bet.grad, bet_r.grad), (bias.grad, bias_r.grad)):
        assert rel_err(got, want) < gt
    ops.RT.new_step(77)
    with torch.no_grad():
        y2, z2 = ops.bias_dropout_residual_prenorm(x, bias, res, gam, bet, 1e-5, p, training=True)
    assert torch.equal(y2, y.detach()) and torch.equal(z2, z.detach())
    # only z used downstream: LayerNorm contributes nothing, the stream gradient passes (masked) to x and unchanged to residual
    xin2, rin2 = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    ops.RT.new_step(77)
    _, z3 = ops.bias_dropout_residual_prenorm(xin2, bias, rin2, gam, bet, 1e-5, p, training=True)
    (z3.float() * dz).sum().backward()
    assert rel_err(rin2.grad, dz.to(dtype)) < 1e-6 if dtype == torch.float32 else rel_err(rin2.grad, dz) < 1e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("two", [True, False])
def test_layernorm_with_post_terms_equals_the_separate_adds(ops, dtype, two):
    """bevbert_layernorm_post_fwd: (LN(x + bias) + post1) + post2 in the LayerNorm's launch -- the embedding
    compositions of vilmodel.py:494-532 / 589-593.  fp32: bit-equal to the separate element-wise adds (same order);
    bf16: one rounding instead of three.  Gradients: the post terms receive dy, the rest is the plain LayerNorm's."""
    rows, H = 333, 768
    g0 = torch.Generator(device="cpu").manual_seed(3)
    mk = lambda *s: torch.randn(*s, generator=g0).to(DEV)
    x, p1, p2 = (mk(rows, H).to(dtype) for _ in range(3))
    bias, gamma, beta = mk(H), mk(H), mk(H)
    leaves = lambda: [t.clone().requires_grad_(True) for t in (x, bias, gamma, beta, p1, p2)]
    a = leaves()
    y = ops.bias_layernorm_plus(a[0].clone(), a[1], a[2], a[3], 1e-12, a[4], a[5] if two else None)
    b = leaves()
    ref = ops.bias_dropout_residual_layernorm(b[0].clone(), b[1], None, b[2], b[3], 1e-12) + b[4]
    if two:
        ref = ref + b[5]
    if dtype == torch.float32:
        assert torch.equal(y, ref)
    else:
        assert float((y.float() - ref.float()).abs().max()) <= 2 ** -6 * float(ref.float().abs().max())
    dy = mk(rows, H).to(dtype)
    y.backward(dy)
    ref.backward(dy)
    for i, name in enumerate(("x", "bias", "gamma", "beta", "post1", "post2")):
        if name == "post2" and not two:
            assert a[i].grad is None
            continue
        assert rel_err(a[i].grad, b[i].grad.float()) < (1e-6 if dtype == torch.float32 else 2e-2), name
    assert torch.equal(a[4].grad, dy)


@pytest.mark.parametrize("eps", [1e-12, 1e-5])
def test_plain_layernorm_matches_torch(ops, eps):
    torch.manual_seed(1)
    x = torch.randn(4, 37, 768, device=DEV, requires_grad=True)
    g = torch.randn(768, device=DEV, requires_grad=True)
    b = torch.randn(768, device=DEV, requires_grad=True)
    y = ops.layernorm(x, g, b, eps)
    x2, g2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, g, b))
    y2 = torch.nn.functional.layer_norm(x2, (768,), g2, b2, eps)
    assert float((y - y2).abs().max()) < 1e-4
    dy = torch.randn_like(y)
    y.backward(dy)
    y2.backward(dy)
    assert rel_err(x.grad, x2.grad) < 1e-4 and rel_err(g.grad, g2.grad) < 1e-4 and rel_err(b.grad, b2.grad) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bias_gelu_fwd_bwd(ops, dtype):
    torch.manual_seed(2)
    rows, C = 333, 3072
    x = (2 * torch.randn(rows, C, device=DEV)).to(dtype).requires_grad_(True)
    bias = torch.randn(C, device=DEV, requires_grad=True)
    y = ops.bias_gelu(x, bias)
    xr = x.detach().float().requires_grad_(True)
    br = bias.detach().clone().requires_grad_(True)
    yr = R.gelu_erf(xr + br)
    assert float(((y.float() - yr).abs() / (1 + yr.abs())).max()) < (1e-5 if dtype == torch.float32 else 8e-3)
    dy = torch.randn(rows, C, device=DEV).to(dtype)
    y.backward(dy)
    yr.backward(dy.float())
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(x.grad, xr.grad) < tol
    assert rel_err(bias.grad, br.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(1280, 768), (64, 768), (4608, 768), (7, 40)])
def test_bias_relu_fwd_bwd(ops, dtype, rows, C):
    """The prediction heads' Linear -> ReLU (pretrain_cmt.py:34-71) with the Linear's bias on the activation kernel:
    relu(x + bias) and its backward (dx = dy where x + bias > 0; the bias gradient is the column sum of dx)."""
    torch.manual_seed(rows + C)
    x = torch.randn(rows, C, device=DEV).to(dtype).requires_grad_(True)
    bias = (0.3 * torch.randn(C, device=DEV)).requires_grad_(True)
    y = ops.bias_relu(x, bias)
    xr = x.detach().float().requires_grad_(True)
    br = bias.detach().clone().requires_grad_(True)
    yr = torch.relu(xr + br)
    assert float((y.float() - yr).abs().max()) <= (0.0 if dtype == torch.float32 else 2 ** -8 * float(yr.abs().max()))
    assert torch.equal(y == 0, yr == 0)
    dy = torch.randn(rows, C, device=DEV).to(dtype)
    y.backward(dy)
    yr.backward(dy.float())
    assert torch.equal(x.grad.float(), xr.grad.to(dtype).float())           # a selection: exact in either dtype
    assert rel_err(bias.grad, br.grad) < (1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_take_rows_gather_and_deterministic_scatter(ops, dtype):
    """ops.take_rows (index_select of activation rows + its backward): duplicates -- two candidate views in one BEV cell,
    the zero-weight padding rows of a static batch that all point at row 0 -- are summed in row order without atomics,
    bit-reproducibly; a second selection from the same tensor (the centre cell) lands in the same gradient tensor."""
    torch.manual_seed(5)
    N, H = 28224, 768
    x = torch.randn(N, H, device=DEV).to(dtype).requires_grad_(True)
    idx = torch.randint(0, N, (4608,), device=DEV)
    idx[100:140] = idx[7]                                   # 41 duplicates of one row
    idx[-300:] = 0                                          # padding rows
    idx2 = torch.cat([idx[:5], torch.randint(0, N, (59,), device=DEV)])      # overlaps the first selection
    a, b = ops.take_rows(x, idx, idx2)
    assert torch.equal(a, x.detach()[idx]) and torch.equal(b, x.detach()[idx2])
    da, db = torch.randn_like(a), torch.randn_like(b)
    (a * da).sum().backward(retain_graph=True)
    g1 = x.grad.clone()
    x.grad = None
    ((a * da).sum() + (b * db).sum()).backward()
    g2 = x.grad.clone()
    ref1 = torch.zeros(N, H, device=DEV, dtype=torch.float64).index_add_(0, idx, da.double())
    ref2 = ref1.clone().index_add_(0, idx2, db.double())
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(g1, ref1) < tol and rel_err(g2, ref2) < tol
    untouched = torch.ones(N, dtype=torch.bool, device=DEV)
    untouched[idx] = False
    untouched[idx2] = False
    assert float(g2[untouched].abs().max()) == 0.0
    # bit-reproducible
    x.grad = None
    a2, b2 = ops.take_rows(x, idx, idx2)
    ((a2 * da).sum() + (b2 * db).sum()).backward()
    assert torch.equal(x.grad, g2)
    # one selection, other widths
    y = torch.randn(50, 40, device=DEV).to(dtype).requires_grad_(True)
    ii = torch.tensor([3, 3, 49, 0, 3], device=DEV)
    r = ops.take_rows(y, ii)
    assert torch.equal(r, y.detach()[ii])
    r.sum().backward()
    assert float(y.grad[3].float().mean()) == 3.0 and float(y.grad[1].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,K,T", [(1280, 7, 100), (28224, 10, 2), (11520, 7, 3), (13, 10, 2), (4097, 12, 5)])
def test_smallk_linear_layernorm_plus(ops, dtype, rows, K, T):
    """smallk.hip: (LayerNorm(feat W^T + b) + post1) + table[idx] with the K <= 16 projection recomputed inside the row
    kernels (vilmodel.py:507-518, 577-583, 589-593), against the fp32 torch composition: outputs, and the gradients the
    backward adds into the arena (weight, bias, gamma, beta, table) and returns (post1).  Deterministic: a second backward
    gives the same bits."""
    from vln_bevbert_amd.arena import ParamArena
    torch.manual_seed(rows + K)
    H = 768

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin, self.ln, self.emb = torch.nn.Linear(K, H), torch.nn.LayerNorm(H, eps=1e-12), torch.nn.Embedding(T, H)
    m = M()
    with torch.no_grad():
        m.ln.weight.uniform_(0.5, 1.5); m.ln.bias.normal_(0, 0.3)
    ref = {n: p.detach().clone().to(DEV).requires_grad_(True) for n, p in m.named_parameters()}
    arena = ParamArena(m, DEV, dtype)
    feat = torch.randn(rows, K, device=DEV)
    post1 = torch.randn(rows, H, device=DEV).to(dtype).requires_grad_(True)
    idx = torch.randint(0, T, (rows,), device=DEV)
    assert ops.smallk_linear_layernorm_plus_supported(feat, m.lin, m.ln, post1, m.emb)
    dy = torch.randn(rows, H, device=DEV).to(dtype)

    def run():
        arena.grads.zero_()
        post1.grad = None
        ops.RT.scratch.reset()
        y = ops.smallk_linear_layernorm_plus(feat, m.lin, m.ln, 1e-12, post1, m.emb, idx)
        y.backward(dy)
        arena.sync()
        torch.cuda.synchronize()
        return y.detach().clone(), arena.grads.clone(), post1.grad.clone()
    y, g, dp = run()
    # reference composition (fp32; the table rows in the compute dtype, as the kernel reads them)
    tbl = ref["emb.weight"].to(dtype).float() if dtype != torch.float32 else ref["emb.weight"]
    z = feat @ ref["lin.weight"].t() + ref["lin.bias"]
    yr = torch.nn.functional.layer_norm(z, (H,), ref["ln.weight"], ref["ln.bias"], 1e-12) + post1.detach().float() + tbl[idx]
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert float((y.float() - yr).abs().max()) < tol * max(1.0, float(yr.abs().max()))
    yr.backward(dy.float())
    gt = 2e-4 if dtype == torch.float32 else 2e-2
    for n in ("lin.weight", "lin.bias", "ln.weight", "ln.bias", "emb.weight"):
        o, k = arena.slices[n]
        got = g[o:o + k].view(ref[n].shape)
        assert rel_err(got, ref[n].grad) < gt, (n, rel_err(got, ref[n].grad))
    assert torch.equal(dp, dy)                                   # d post1 = dy
    y2, g2, _ = run()
    assert torch.equal(y2, y) and torch.equal(g2, g)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_semantic_head_loss_select_and_bce(ops, dtype):
    """ops.sem_select + ops.bce_rows + ops.weighted_mean (pretrain_cmt.py:391-441 on a static batch): the cells with both
    masks set, compacted in ascending order into a fixed capacity (padding row 0 / weight 0), the divisor count x classes,
    the per-row BCE-with-logits sums read through the index, and the weighted mean -- against the torch composition,
    forward and gradient; also an empty selection and one that exceeds the capacity."""
    torch.manual_seed(9)
    N, C, cap = 28224, 40, 4608
    m1 = torch.rand(N, device=DEV) < 0.4
    m2 = torch.rand(N, device=DEV) < 0.35
    labels = (torch.rand(N, C, device=DEV) < 0.1).to(torch.uint8)
    idx, valid, denom = ops.sem_select(m1, m2, cap, C)
    want = torch.nonzero(m1 & m2).squeeze(1)
    n = int(want.numel())
    assert 0 < n < cap and torch.equal(idx[:n], want) and int(idx[n:].abs().max()) == 0
    assert float(valid.sum()) == n and bool((valid[:n] == 1).all()) and float(denom) == n * C
    logits = (2 * torch.randn(cap, C, device=DEV)).to(dtype).requires_grad_(True)
    loss = ops.weighted_mean(ops.bce_rows(logits, labels, idx), valid, denom)
    lr = logits.detach().float().requires_grad_(True)
    per = torch.nn.functional.binary_cross_entropy_with_logits(lr, labels[idx].float(), reduction="none")
    ref = (per * valid[:, None]).sum() / (n * C)
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    loss.backward()
    ref.backward()
    assert rel_err(logits.grad, lr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)
    assert float(logits.grad[n:].float().abs().max()) == 0.0            # padding rows carry no gradient
    # one mask only; nothing selected; more cells than the capacity (the first ``cap`` in order, the divisor still the count)
    idx1, valid1, denom1 = ops.sem_select(m1, None, N, C)
    assert torch.equal(idx1[:int(m1.sum())], torch.nonzero(m1).squeeze(1)) and float(denom1) == float(m1.sum()) * C
    idx0, valid0, denom0 = ops.sem_select(torch.zeros(N, dtype=torch.bool, device=DEV), None, 64, C)
    assert float(valid0.sum()) == 0 and float(denom0) == 0 and int(idx0.abs().max()) == 0
    idx2, valid2, denom2 = ops.sem_select(torch.ones(N, dtype=torch.bool, device=DEV), None, 100, C)
    assert torch.equal(idx2, torch.arange(100, device=DEV)) and float(valid2.sum()) == 100 and float(denom2) == N * C


def test_embed_sum_layernorm(ops):
    torch.manual_seed(3)
    V, H, B, L = 500, 768, 3, 17
    word = torch.randn(V, H, device=DEV, requires_grad=True)
    pos = torch.randn(64, H, device=DEV, requires_grad=True)
    typ = torch.randn(2, H, device=DEV, requires_grad=True)
    g = torch.randn(H, device=DEV, requires_grad=True)
    b = torch.randn(H, device=DEV, requires_grad=True)
    ids = torch.randint(0, V, (B, L), device=DEV)
    ids[0, :3] = 7                       # repeated ids: the gradient must accumulate
    y = ops.embed_sum_layernorm(ids, word, pos, typ, g, b, 1e-12, 0)
    ref_in = [t.detach().clone().requires_grad_(True) for t in (word, pos, typ, g, b)]
    w2, p2, t2, g2, b2 = ref_in
    yr = torch.nn.functional.layer_norm(w2[ids] + p2[:L][None] + t2[0], (H,), g2, b2, 1e-12)
    assert float((y - yr).abs().max()) < 1e-4
    dy = torch.randn_like(y)
    y.backward(dy)
    yr.backward(dy)
    for a_, r_ in zip((word, pos, typ, g, b), ref_in):
        assert rel_err(a_.grad, r_.grad) < 1e-4


@pytest.mark.parametrize("table_rows,rows,dtype", [(2, 28224, torch.bfloat16), (3, 11520, torch.bfloat16),
                                                   (100, 1280, torch.bfloat16), (2, 1, torch.float32),
                                                   (5, 77, torch.float32)])
def test_small_table_embedding_grad_sliced(ops, table_rows, rows, dtype):
    """vilmodel.py nav_type / step-id embedding backward: sliced partial sums + batched fold == index_add_, and the same
    bits on every run (no atomics)."""
    torch.manual_seed(5)
    H = 768
    ids = torch.randint(0, table_rows, (rows,), device=DEV)
    d = torch.randn(rows, H, device=DEV).to(dtype)
    want = torch.zeros(table_rows, H, device=DEV, dtype=torch.float64).index_add_(0, ids, d.double()) + 1.0
    outs = []
    for _ in range(2):
        sink = torch.ones(table_rows, H, device=DEV)          # accumulate semantics: sink += ...
        ops.embedding_grad_small(ids, d, sink, table_rows)
        ops.WgradStream.flush_all()
        torch.cuda.synchronize()
        outs.append(sink)
    assert float((outs[0].double() - want).abs().max() / want.abs().max()) < 1e-5
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("rows,V,H,dtype", [(5120, 30522, 768, torch.bfloat16), (5120, 9, 768, torch.bfloat16),
                                            (3000, 40, 64, torch.float32), (51, 500, 768, torch.float32),
                                            (1, 3, 128, torch.float32)])
def test_word_embedding_grad_is_deterministic_and_honours_padding_idx(ops, rows, V, H, dtype):
    """bevbert_embedding_grad (backward of BertEmbeddings' word lookup, vilmodel.py:50,67): equals index_add_ with the
    padding row left alone, accumulates into the sink, and gives the SAME BITS on every run -- no atomics.  (5120, 9):
    ~570 rows per id; (3000, 40) with 1 700 rows of one id spread over the whole range: every wave of the leader finds
    hundreds of matches in its quarter."""
    from vln_bevbert_amd.lib import call, dtype_code, ptr, stream
    torch.manual_seed(11)
    ids = torch.randint(0, V, (rows,), device=DEV)
    if rows == 3000:
        ids[torch.randperm(rows, device=DEV)[:1700]] = 5
    d = torch.randn(rows, H, device=DEV).to(dtype)
    for pad in (-1, 0):
        keep = ids != pad
        want = torch.zeros(V, H, device=DEV, dtype=torch.float64).index_add_(0, ids[keep], d[keep].double()) + 1.0
        outs = []
        for _ in range(3):
            sink = torch.ones(V, H, device=DEV)
            call("bevbert_embedding_grad", ptr(ids), ptr(d), ptr(sink), rows, H, pad, dtype_code(d), stream())
            torch.cuda.synchronize()
            outs.append(sink)
        err = float((outs[0].double() - want).abs().max() / want.abs().max())
        assert err < 1e-5, (pad, err)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        if pad == 0:
            assert torch.equal(outs[0][0], torch.ones(H, device=DEV))


def test_graph_bias_gradient_is_deterministic(ops):
    """The additive graph bias (B, G, G) is shared by the 12 heads: the backward kernels store per-head gradients and the
    host folds the heads -- the same bits on every run, for the exact and the MFMA kernels."""
    for dtype, impl in ((torch.float32, 1), (torch.bfloat16, 2)):
        q, k, v, km, bias, nh = _make_attn_inputs(5, 23, 23, "neg", True, dtype, seed=4)
        outs = []
        for _ in range(3):
            bi = bias.clone().requires_grad_(True)
            qi, ki, vi = (t.clone().requires_grad_(True) for t in (q, k, v))
            o = ops.attention(qi, ki, vi, km, bi, nh, impl=impl)
            o.backward(torch.ones_like(o) * 0.01 + o.detach() * 0.1)
            torch.cuda.synchronize()
            outs.append(bi.grad.clone())
        assert float(outs[0].abs().sum()) > 0
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graph_bias_shared_by_the_layers_of_an_encoder(ops, dtype):
    """ops.graph_bias (sprel_linear over the pairwise node distances, vilmodel.py:575-577): one forward launch hands a view
    to every x-layer; each layer's attention backward leaves its per-head bias gradients in a shared buffer and ONE launch
    reduces them to d sprel_linear.weight / .bias in the arena -- against the torch composition through the fp32 attention
    reference, for three layers."""
    from vln_bevbert_amd.arena import ParamArena
    torch.manual_seed(4)
    B, G, nh, L = 5, 20, 12, 3

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.sprel_linear = torch.nn.Linear(1, 1)
    m = M()
    w0, b0 = float(m.sprel_linear.weight), float(m.sprel_linear.bias)
    arena = ParamArena(m, DEV, torch.float32)
    dists = torch.rand(B, G, G, device=DEV) * 3
    km = torch.zeros(B, G, device=DEV)
    km[1, 15:] = -10000.0
    qkv = [tuple(torch.randn(B, G, 768, device=DEV).to(dtype) for _ in range(3)) for _ in range(L)]
    do = [torch.randn(B, G, 768, device=DEV).to(dtype) for _ in range(L)]
    biases = ops.graph_bias(dists, m.sprel_linear.weight, m.sprel_linear.bias, L, nh)
    assert len(biases) == L and all(float((x - (dists * w0 + b0)).abs().max()) < 1e-6 for x in biases)      # (the kernel fuses the multiply-add)

    def run():
        arena.grads.zero_()
        bs = ops.graph_bias(dists, m.sprel_linear.weight, m.sprel_linear.bias, L, nh)
        tot = 0
        for (q, k, v), bi, d in zip(qkv, bs, do):
            tot = tot + (ops.attention(q, k, v, km, bi, nh, impl=2 if dtype == torch.bfloat16 else 1).float() * d.float()).sum()
        tot.backward()
        torch.cuda.synchronize()
        return arena.grads.clone()
    g = run()
    wr = torch.tensor(w0, device=DEV, requires_grad=True)
    br = torch.tensor(b0, device=DEV, requires_grad=True)
    tot = 0
    for (q, k, v), d in zip(qkv, do):
        tot = tot + (_attn_ref(q.float(), k.float(), v.float(), km, dists * wr + br, nh) * d.float()).sum()
    tot.backward()
    ow, _ = arena.slices["sprel_linear.weight"]
    ob, _ = arena.slices["sprel_linear.bias"]
    tol = 1e-3 if dtype == torch.float32 else 5e-2
    assert abs(float(g[ow]) - float(wr.grad)) < tol * max(1.0, abs(float(wr.grad))), (float(g[ow]), float(wr.grad))
    assert abs(float(g[ob]) - float(br.grad)) < tol * max(1.0, abs(float(wr.grad))), (float(g[ob]), float(br.grad))
    assert torch.equal(run(), g)                           # fixed summation order


@pytest.mark.parametrize("dtype,fuse", [(torch.float32, True), (torch.float32, False), (torch.bfloat16, True)])
def test_fused_sap_loss_tail_matches_the_torch_composition(ops, dtype, fuse):
    """ops.sap_loss == pretrain_cmt.forward_sap's tail written with torch ops (masked fills, fuse_sap_logits, three
    cross-entropies): loss per sample and the gradients w.r.t. the three head outputs."""
    import torch.nn.functional as F
    from vln_bevbert_amd.pretrain_cmt import fuse_sap_logits
    torch.manual_seed(9)
    B, G, K, P = 7, 23, 6, 441
    graw = torch.randn(B, G, device=DEV).to(dtype).requires_grad_(True)
    lraw = torch.randn(B, K, device=DEV).to(dtype).requires_grad_(True)
    fraw = torch.randn(B, 1, device=DEV).to(dtype).requires_grad_(True) if fuse else None
    gmap_lens = torch.randint(4, G + 1, (B,), device=DEV)
    visited = torch.rand(B, G, device=DEV) < 0.3
    visited[:, 1] = False                                   # the labelled node stays selectable
    nav_masks = torch.rand(B, P, device=DEV) < 0.7
    cand_idxs = torch.randint(0, P, (B, K), device=DEV)
    nav_masks[torch.arange(B, device=DEV), cand_idxs[:, 2]] = True
    src = torch.randint(0, K + 2, (B, G), device=DEV)       # K: "sum over visited candidates", K + 1: nothing
    src[:, 1] = K + 1                                       # (the labelled node must not inherit a masked candidate's -inf)
    vis_c = torch.rand(B, K, device=DEV) < 0.4
    vis_c &= nav_masks[torch.arange(B, device=DEV)[:, None], cand_idxs]      # (a masked visited candidate gives -inf)
    gl_lab = torch.ones(B, dtype=torch.long, device=DEV)
    ll_lab = torch.full((B,), 2, dtype=torch.long, device=DEV)
    w = torch.randn(B, device=DEV)

    loss = ops.sap_loss(graw, lraw, fraw, visited, gmap_lens, nav_masks, cand_idxs, src, vis_c, gl_lab, ll_lab)
    (loss * w).sum().backward()
    got = [loss.detach().clone()] + [t.grad.clone() for t in (graw, lraw) + ((fraw,) if fuse else ())]

    ref_in = [t.detach().float().clone().requires_grad_(True) for t in (graw, lraw) + ((fraw,) if fuse else ())]
    g2, l2 = ref_in[0], ref_in[1]
    fw = torch.sigmoid(ref_in[2]) if fuse else 0.5
    gl = (g2 * fw).masked_fill(visited, -float("inf"))
    gl = gl.masked_fill(torch.arange(G, device=DEV)[None] >= gmap_lens[:, None], -float("inf"))
    cm = nav_masks[torch.arange(B, device=DEV)[:, None], cand_idxs]
    ll = (l2 * (1 - fw)).masked_fill(cm.logical_not(), -float("inf"))
    fu = fuse_sap_logits(gl, ll, src, vis_c)
    want = F.cross_entropy(gl, gl_lab, reduction="none") + F.cross_entropy(ll, ll_lab, reduction="none") \
        + F.cross_entropy(fu, gl_lab, reduction="none")
    (want * w).sum().backward()
    ref = [want.detach()] + [t.grad for t in ref_in]
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert torch.isfinite(want).all()
    for a_, r_ in zip(got, ref):
        assert float((a_.float() - r_).abs().max()) <= tol * max(1.0, float(r_.abs().max())), (a_, r_)


@pytest.mark.parametrize("rows,C,dtype", [(5, 30522, torch.bfloat16), (3, 30522, torch.float32), (4, 250002, torch.bfloat16),
                                          (7, 41, torch.float32), (2, 4, torch.bfloat16)])
def test_cross_entropy_rows_matches_torch(ops, rows, C, dtype):
    """The MLM head's loss: F.cross_entropy(logits.float(), target, reduction='none') and its gradient, read straight
    from the logits' dtype; vocabulary sizes that are not multiples of four exercise the ragged row ends."""
    import torch.nn.functional as F
    torch.manual_seed(13)
    x = (3 * torch.randn(rows, C, device=DEV)).to(dtype).requires_grad_(True)
    t = torch.randint(0, C, (rows,), device=DEV)
    t[0] = C - 1
    w = torch.randn(rows, device=DEV)
    loss = ops.cross_entropy_rows(x, t)
    (loss * w).sum().backward()
    xr = x.detach().float().requires_grad_(True)
    want = F.cross_entropy(xr, t, reduction="none")
    (want * w).sum().backward()
    assert float((loss - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    tol = 1e-6 if dtype == torch.float32 else 4e-3          # bf16: one rounding of the fp32 gradient
    assert float((x.grad.float() - xr.grad).abs().max()) <= tol * max(1.0, float(xr.grad.abs().max()))


def test_segment_wsum_matches_oracle_aggregation(ops):
    from vln_bevbert_amd.vilmodel import build_gmap_csr
    cfg = BevBertConfig()
    b = synthetic.make_batch(cfg, "sap", 4, seed=31, ragged=True)
    T, V = b["traj_view_img_fts"].shape[:2]
    torch.manual_seed(4)
    emb = torch.randn(T, V, 768)
    masks = R.seq_mask(b["traj_vp_view_lens"], V)
    want = R.aggregate_gmap(emb, masks, b["traj_vp_view_lens"], b["traj_step_lens"], b["traj_vpids"],
                            b["traj_cand_vpids"], b["gmap_vpids"])
    csr, G = build_gmap_csr(b["traj_step_lens"], b["traj_vp_view_lens"].tolist(), b["traj_vpids"],
                            b["traj_cand_vpids"], b["gmap_vpids"], V, DEV)
    src = emb.reshape(-1, 768).to(DEV).requires_grad_(True)
    got = ops.segment_wsum(src, csr).view(4, G, 768)
    assert float((got.cpu() - want).abs().max()) < 1e-5
    dy = torch.randn_like(got)
    got.backward(dy)
    e2 = emb.clone().requires_grad_(True)
    R.aggregate_gmap(e2, masks, b["traj_vp_view_lens"], b["traj_step_lens"], b["traj_vpids"], b["traj_cand_vpids"],
                     b["gmap_vpids"]).backward(dy.cpu())
    assert float((src.grad.cpu().view_as(e2) - e2.grad).abs().max()) < 1e-5


# ----------------------------------------------------------------------------- K2 attention
def _attn_ref(q, k, v, key_mask, bias, nh, keep=None, p=0.0):
    """fp32 reference = oracle's attention math on already-projected Q/K/V (vilmodel.py:116-137)."""
    B, Lq, H = q.shape
    d = H // nh
    hq = q.view(B, Lq, nh, d).permute(0, 2, 1, 3)
    hk = k.view(B, -1, nh, d).permute(0, 2, 1, 3)
    hv = v.view(B, -1, nh, d).permute(0, 2, 1, 3)
    s = hq @ hk.transpose(-1, -2) / math.sqrt(d)
    if key_mask is not None:
        s = s + key_mask[:, None, None, :]
    if bias is not None:
        s = s + bias[:, None]
    pr = torch.softmax(s, -1)
    if keep is not None:
        pr = torch.where(keep, pr / (1 - p), torch.zeros_like(pr))
    return (pr @ hv).permute(0, 2, 1, 3).reshape(B, Lq, H)


ATTN_CASES = [
    # B, Lq, Lk, mask kind, bias
    (2, 441, 441, None, False),        # BEV self-attention
    (2, 441, 80, "neg", False),        # BEV -> text cross-attention
    (3, 17, 17, "neg", True),          # gmap self-attention with graph bias
    (5, 38, 38, "inf", False),         # panorama encoder, boolean key padding
    (2, 80, 441, None, False),         # MLM: text -> BEV
    (2, 80, 80, "neg", False),         # text self-attention
    (1, 130, 200, "neg", True),        # odd sizes, RxR-length text
]


def _make_attn_inputs(B, Lq, Lk, mask_kind, with_bias, dtype, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    nh, H = 12, 768
    q = torch.randn(B, Lq, H, generator=g).to(DEV).to(dtype)
    k = torch.randn(B, Lk, H, generator=g).to(DEV).to(dtype)
    v = torch.randn(B, Lk, H, generator=g).to(DEV).to(dtype)
    km = None
    if mask_kind is not None:
        lens = torch.randint(max(1, Lk // 2), Lk + 1, (B,), generator=g)
        lens[0] = Lk
        valid = torch.arange(Lk)[None] < lens[:, None]
        val = -10000.0 if mask_kind == "neg" else float("-inf")
        km = torch.zeros(B, Lk).masked_fill(~valid, val).to(DEV)
    bias = (0.5 * torch.randn(B, Lq, Lk, generator=g)).to(DEV) if with_bias else None
    return q, k, v, km, bias, nh


@pytest.mark.parametrize("case", ATTN_CASES)
@pytest.mark.parametrize("impl,dtype", [(1, torch.float32), (2, torch.bfloat16), (3, torch.bfloat16), (1, torch.bfloat16)])
def test_attention_fwd_bwd(ops, case, impl, dtype):
    """impl 1 = exact fp32-arithmetic kernels, 2 = MFMA (single-pass backward where it applies), 3 = MFMA with the
    two-kernel backward (the path of key sequences beyond 448)."""
    B, Lq, Lk, mk, wb = case
    q, k, v, km, bias, nh = _make_attn_inputs(B, Lq, Lk, mk, wb, dtype)
    qi, ki, vi = (t.clone().requires_grad_(True) for t in (q, k, v))
    bi = bias.clone().requires_grad_(True) if wb else None
    o = ops.attention(qi, ki, vi, km, bi, nh, impl=impl)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    br = bias.clone().requires_grad_(True) if wb else None
    orf = _attn_ref(qr, kr, vr, km, br, nh)
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    scale = float(orf.abs().max())
    assert float((o.float() - orf).abs().max()) < tol * max(1.0, scale), "forward"
    do = torch.randn(B, Lq, 768, device=DEV).to(dtype)
    o.backward(do)
    orf.backward(do.float())
    gt = 1e-3 if dtype == torch.float32 else 2.5e-2
    assert rel_err(qi.grad, qr.grad) < gt, "dq"
    assert rel_err(ki.grad, kr.grad) < gt, "dk"
    assert rel_err(vi.grad, vr.grad) < gt, "dv"
    if wb:
        assert rel_err(bi.grad, br.grad) < gt, "dbias"


@pytest.mark.parametrize("Lq,Lk,mk", [(80, 80, "neg"), (441, 80, "neg"), (17, 80, "inf"), (80, 17, "neg"), (36, 36, "inf"),
                                       (100, 96, None), (33, 5, None), (70, 49, "neg")])
@pytest.mark.parametrize("small_fwd", ["1", "0"])
def test_attention_short_key_kernels_with_dropout(ops, Lq, Lk, mk, small_fwd, monkeypatch):
    """attn_small.hip (Lk <= 96, no graph bias): forward (inline hash, one tile set) and the one-wave backward, which
    reads the keep bits the forward left, against the fp32 reference under the exported mask -- every key-tile count
    the launcher instantiates (2, 3, 5, 6) and query counts with partial tiles / partial 32-query chunks.  The forward
    and the one-wave backward are opt-in (they measured no faster than the tiled ones, capi.hip; the library reads the
    switch per call); with queries AND keys up to 96 the backward is the independent-waves kernel either way."""
    monkeypatch.setenv("BEVBERT_ATTN_SMALL", small_fwd)     # "0": tiled forward (keep bits from either) + new backward
    B, p, dtype = 3, 0.1, torch.bfloat16
    q, k, v, km, _, nh = _make_attn_inputs(B, Lq, Lk, mk, False, dtype, seed=Lq + Lk)
    ops.RT.new_step(1234 + Lk)
    qi, ki, vi = (t.clone().requires_grad_(True) for t in (q, k, v))
    o = ops._Attention.apply("sep", qi, ki, vi, km, None, nh, p, 2)
    Lk2 = (Lk + 1) // 2 * 2
    keep = ops.dropout_keep_mask(B * nh * Lq * Lk2, p, ops.RT.seed, 0, DEV).view(B, nh, Lq, Lk2)[..., :Lk]
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    orf = _attn_ref(qr, kr, vr, km, None, nh, keep, p)
    assert float((o.float() - orf).abs().max()) < 1.5e-2 * max(1.0, float(orf.abs().max()))
    do = torch.randn_like(orf).to(dtype)
    o.backward(do)
    orf.backward(do.float())
    for name, a, b_ in (("dq", qi.grad, qr.grad), ("dk", ki.grad, kr.grad), ("dv", vi.grad, vr.grad)):
        assert bool(torch.isfinite(a).all()), name
        assert rel_err(a, b_) < 3e-2, (name, rel_err(a, b_))


@pytest.mark.parametrize("Lq,Lk,mk,p", [(80, 80, "neg", 0.1), (80, 80, None, 0.0), (36, 36, "inf", 0.1), (20, 20, "neg", 0.1), (20, 80, "neg", 0.1),
                                         (80, 20, "inf", 0.1), (96, 96, "neg", 0.1), (1, 2, None, 0.1), (16, 16, None, 0.0), (65, 33, "neg", 0.3),
                                         (5, 96, "neg", 0.1), (95, 7, None, 0.1), (441, 80, "neg", 0.1), (441, 80, "neg", 0.0), (300, 49, None, 0.1)])
def test_attention_short_kernels_of_round_6(ops, Lq, Lk, mk, p, monkeypatch):
    """attn_short.hip: the forward for key sequences up to 96 (any query count; keep bits hashed inline for small score
    matrices, read from the caller's words for the 441 x 80 BEV <- text shape) and the one-pass backward for queries AND
    keys up to 96 -- every tile-count instantiation (2, 3, 5, 6 on both axes), partial tiles, odd tile counts (the zero
    half of the last 32-chunk), a single query against two keys -- against fp32 math under the exported mask; the library must
    say it took these kernels (bevbert_attn_last_path), and with BEVBERT_ATTN_SHORT=0 the kernels of rounds 2-5."""
    from vln_bevbert_amd import lib
    L = lib.load()
    B, dtype = 3, torch.bfloat16
    q, k, v, km, _, nh = _make_attn_inputs(B, Lq, Lk, mk, False, dtype, seed=7 * Lq + Lk)
    Lk2 = (Lk + 1) // 2 * 2
    outs = {}
    for arm in ("1", "0"):
        monkeypatch.setenv("BEVBERT_ATTN_SHORT", arm)
        ops.RT.new_step(4321 + Lq)
        qi, ki, vi = (t.clone().requires_grad_(True) for t in (q, k, v))
        o = ops._Attention.apply("sep", qi, ki, vi, km, None, nh, p, 2)
        path_f = L.bevbert_attn_last_path(0).decode()
        keep = None
        if p > 0:
            keep = ops.dropout_keep_mask(B * nh * Lq * Lk2, p, ops.RT.seed, 0, DEV).view(B, nh, Lq, Lk2)[..., :Lk]
        qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
        orf = _attn_ref(qr, kr, vr, km, None, nh, keep, p)
        assert float((o.float() - orf).abs().max()) < 1.5e-2 * max(1.0, float(orf.abs().max())), (arm, path_f)
        do = torch.randn_like(orf).to(dtype)
        o.backward(do)
        torch.cuda.synchronize()
        orf.backward(do.float())
        for name, a, b_ in (("dq", qi.grad, qr.grad), ("dk", ki.grad, kr.grad), ("dv", vi.grad, vr.grad)):
            assert bool(torch.isfinite(a).all()), (arm, name)
            assert rel_err(a, b_) < 3e-2, (arm, name, rel_err(a, b_))
        outs[arm] = (path_f, o.detach().clone(), qi.grad.clone())
    assert outs["1"][0] == "attn_short_fwd" and outs["0"][0] != "attn_short_fwd", (outs["1"][0], outs["0"][0])
    # the two generations agree far inside the bf16 gate (same mask, same arithmetic order up to the tiling)
    assert float((outs["1"][1].float() - outs["0"][1].float()).abs().max()) < 2e-2 * max(1.0, float(outs["0"][1].float().abs().max()))


def test_attention_short_backward_is_the_dispatched_kernel_and_leaves_other_rows_alone(ops):
    """The backward dispatch (thread-local path record is read on the thread that issued the call) and the packed-QKV
    strides the model uses: gradients land in their column slices of the packed buffers, nothing else is written."""
    from vln_bevbert_amd import lib
    from vln_bevbert_amd.lib import call, dtype_code, ptr, stream
    import math
    L = lib.load()
    B, Lq, nh, H, p = 4, 80, 12, 768, 0.1
    torch.manual_seed(3)
    qkv = torch.randn(B, Lq, 3 * H, device=DEV).bfloat16()
    q, k, v = qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:]
    o = torch.empty(B, Lq, H, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, nh, Lq, device=DEV)
    bits = torch.zeros(ops._drop_bits_words(B, nh, Lq, Lq), dtype=torch.int64, device=DEV)
    st = ops._strides(q, k, v, o)
    call("bevbert_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), None, None, st, B, nh, Lq, Lq, 64, 0.125, dtype_code(q), 0,
         p, 11, 0, ptr(bits), 0, stream())
    assert L.bevbert_attn_last_path(0) == b"attn_short_fwd"
    do = torch.randn_like(o)
    dqkv = torch.full_like(qkv, 7.0)
    delta = torch.empty_like(lse)
    stg = ops._strides(q, k, v, o)
    # gradient buffers share the operand strides (include/bevbert_hip.h): the packed layout of the fused QKV projection
    dq2, dk2, dv2 = dqkv[..., :H], dqkv[..., H:2 * H], dqkv[..., 2 * H:]
    call("bevbert_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq2), ptr(dk2), ptr(dv2), None, None,
         None, stg, B, nh, Lq, Lq, 64, 0.125, dtype_code(q), 0, p, 11, 0, ptr(bits), stream())
    assert L.bevbert_attn_last_path(1) == b"attn_short_bwd"
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dqkv.float()).all()) and not bool((dqkv == 7.0).any())     # every element written exactly where it belongs
    Lk2 = Lq
    keep = ops.dropout_keep_mask(B * nh * Lq * Lk2, p, 11, 0, DEV).view(B, nh, Lq, Lk2)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    orf = _attn_ref(qr, kr, vr, None, None, nh, keep, p)
    orf.backward(do.float())
    assert rel_err(dq2, qr.grad) < 3e-2 and rel_err(dk2, kr.grad) < 3e-2 and rel_err(dv2, vr.grad) < 3e-2


@pytest.mark.parametrize("B,Lq,Lk,mk,p,wgs", [(2, 441, 441, None, 0.1, "5"), (2, 441, 441, "neg", 0.0, "24"), (1, 300, 290, "inf", 0.1, "3"),
                                             (3, 500, 448, "neg", 0.1, "7"), (2, 257, 385, None, 0.1, "1")])
def test_attention_persistent_forward_of_the_long_shapes(ops, B, Lq, Lk, mk, p, wgs, monkeypatch):
    """attn_fwd4.hip (no bias, 256 < Lk <= 448, Lq > 256): one workgroup per CU walking (batch, head, 448-query block) items,
    a producer wave streaming the K / V tiles across item boundaries, keep bits in the per-lane layout.  Forced on for small
    batches (the launcher takes it only when the items fill at least two rounds of CUs) with a grid of `wgs` workgroups, so
    that a workgroup walks several items (uneven counts included) and more than one query block per head (Lq = 500);
    against the fp32 reference under the exported mask AND against the 4-wave kernel on the same inputs."""
    dtype = torch.bfloat16
    q, k, v, km, _, nh = _make_attn_inputs(B, Lq, Lk, mk, False, dtype, seed=Lq + Lk)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BEVBERT_ATTN_FWD4", mode)
        monkeypatch.setenv("BEVBERT_FWD4_WGS", wgs)
        ops.RT.new_step(4321 + Lk)
        qi, ki, vi = (t.clone().requires_grad_(True) for t in (q, k, v))
        o = ops._Attention.apply("sep", qi, ki, vi, km, None, nh, p, 2)
        do = torch.randn(o.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)).to(dtype)
        o.backward(do)
        outs[mode] = (o.detach().float(), qi.grad.float(), ki.grad.float(), vi.grad.float())
    Lk2 = (Lk + 1) // 2 * 2
    keep = None
    if p > 0:
        keep = ops.dropout_keep_mask(B * nh * Lq * Lk2, p, ops.RT.seed, 0, DEV).view(B, nh, Lq, Lk2)[..., :Lk]
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    orf = _attn_ref(qr, kr, vr, km, None, nh, keep, p)
    scale = max(1.0, float(orf.abs().max()))
    assert bool(torch.isfinite(outs["1"][0]).all())
    assert float((outs["1"][0] - orf).abs().max()) < 1.5e-2 * scale
    # same products in the same order as the 4-wave kernel; only the row sums are accumulated in a different order
    assert float((outs["1"][0] - outs["0"][0]).abs().max()) < 2 ** -7 * scale
    orf.backward(do.float())
    for name, a, b_ in zip(("dq", "dk", "dv"), outs["1"][1:], (qr.grad, kr.grad, vr.grad)):
        assert rel_err(a, b_) < 3e-2, (name, rel_err(a, b_))       # the backward reads the log-sum-exp this forward wrote


@pytest.mark.parametrize("Lk", [140, 441, 36])
@pytest.mark.parametrize("impl,dtype", [(1, torch.float32), (2, torch.bfloat16), (3, torch.bfloat16)])
def test_attention_dropout_matches_exported_mask(ops, impl, dtype, Lk):
    B, Lq, p = 2, 100, 0.1
    q, k, v, km, _, nh = _make_attn_inputs(B, Lq, Lk, "neg", False, dtype, seed=5)
    ops.RT.new_step(99)
    qi, ki, vi = (t.clone().requires_grad_(True) for t in (q, k, v))
    o = ops._Attention.apply("sep", qi, ki, vi, km, None, nh, p, impl)
    Lk2 = (Lk + 1) // 2 * 2            # the kernels index dropout elements with the key count rounded up to even
    keep = ops.dropout_keep_mask(B * nh * Lq * Lk2, p, ops.RT.seed, 0, DEV).view(B, nh, Lq, Lk2)[..., :Lk]
    assert abs(float(keep.float().mean()) - 0.9) < 0.01
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    orf = _attn_ref(qr, kr, vr, km, None, nh, keep, p)
    tol = 1e-4 if dtype == torch.float32 else 1.5e-2
    assert float((o.float() - orf).abs().max()) < tol * max(1.0, float(orf.abs().max()))
    do = torch.randn_like(orf).to(dtype)
    o.backward(do)
    orf.backward(do.float())
    gt = 1e-3 if dtype == torch.float32 else 3e-2
    assert rel_err(qi.grad, qr.grad) < gt and rel_err(ki.grad, kr.grad) < gt and rel_err(vi.grad, vr.grad) < gt


@pytest.mark.parametrize("B,Lq,Lk,mk,wb,p", [(3, 441, 441, None, False, 0.1), (2, 441, 80, "neg", False, 0.1),
                                            (3, 17, 17, "neg", True, 0.1), (2, 80, 441, None, False, 0.0),
                                            (2, 65, 129, "neg", False, 0.1), (2, 200, 448, "neg", False, 0.1)])
def test_attention_single_pass_backward_equals_two_kernel_backward(ops, B, Lq, Lk, mk, wb, p):
    """attn_bwd1.hip (one workgroup per (batch, head), dQ without atomics, keep bits from the forward) against the
    two-kernel backward on the same forward, dropout ON: the same mask, the same softmax statistics -> gradients agree
    to bf16 rounding of dS / P (both pack them to bf16 for the second contraction)."""
    q, k, v, km, bias, nh = _make_attn_inputs(B, Lq, Lk, mk, wb, torch.bfloat16, seed=11)
    do = torch.randn(B, Lq, 768, device=DEV).bfloat16()
    grads = {}
    for impl in (2, 3):
        ops.RT.new_step(4321)
        qi, ki, vi = (t.clone().requires_grad_(True) for t in (q, k, v))
        bi = bias.clone().requires_grad_(True) if wb else None
        o = ops._Attention.apply("sep", qi, ki, vi, km, bi, nh, p, impl)
        o.backward(do)
        grads[impl] = (o.detach().float(), qi.grad.float(), ki.grad.float(), vi.grad.float(),
                       bi.grad if wb else torch.zeros(1, device=DEV))
    assert torch.equal(grads[2][0], grads[3][0])                  # one forward kernel
    for name, a, b_ in zip(("dq", "dk", "dv", "dbias"), grads[2][1:], grads[3][1:]):
        assert bool(torch.isfinite(a).all()), name
        assert rel_err(a, b_) < 1e-2, (name, rel_err(a, b_))


@pytest.mark.parametrize("B,Lq,Lk", [(2, 441, 441), (1, 100, 140), (2, 80, 36), (1, 33, 448)])
def test_attention_keep_bit_workspace_holds_the_exported_mask_in_both_layouts(ops, B, Lq, Lk):
    """bevbert_attn_drop_bits: the forward-layout words (bit l = query 16 q16 + (l & 15), key 64 k64 + 16 t + 4 (l >> 4) + r)
    and the backward-layout words (bit l = query 32 q32 + 16 tt + 4 (l >> 4) + r, key 64 k64 + 16 t + (l & 15)) both decode to
    the mask of the element-indexed stream every other dropout consumer uses."""
    nh, p = 12, 0.1
    ops.RT.new_step(77)
    bits = ops.attn_drop_bits(B, nh, Lq, Lk, p, ops.RT.seed, 5, DEV).cpu().numpy().view(np.uint64)
    Lk2 = (Lk + 1) // 2 * 2
    keep = ops.dropout_keep_mask(B * nh * Lq * Lk2, p, ops.RT.seed, 5, DEV).view(B * nh, Lq, Lk2).cpu().numpy()
    nq16, nk64 = (Lq + 127) // 128 * 8, (Lk + 63) // 64
    half = B * nh * nq16 * nk64 * 16
    assert bits.shape[0] == 3 * half
    lanes = np.arange(64)
    f = bits[:half].reshape(B * nh, nq16, nk64, 4, 4)
    q = np.arange(nq16)[:, None, None, None, None] * 16 + (lanes & 15)
    k = (np.arange(nk64)[None, :, None, None, None] * 64 + np.arange(4)[None, None, :, None, None] * 16
         + (lanes >> 4) * 4 + np.arange(4)[None, None, None, :, None])
    got = ((f[..., None] >> lanes.astype(np.uint64)) & np.uint64(1)).astype(bool)         # (bh, q16, k64, t, r, lane)
    ok = (q < Lq) & (k < Lk)
    qq, kk = np.broadcast_arrays(np.minimum(q, Lq - 1), np.minimum(k, Lk - 1))
    want = keep[:, qq, kk]
    assert np.array_equal(got[:, ok], want[:, ok]), "forward layout"
    if not 256 < Lk <= 448:          # the backward layout is only produced for the shapes the 7+1-wave backward takes
        return
    bw = bits[half:2 * half].reshape(B * nh, nq16 // 2, nk64, 2, 4, 4)                   # (bh, q32, k64, tt, t, r)
    q = (np.arange(nq16 // 2)[:, None, None, None, None, None] * 32 + np.arange(2)[None, None, :, None, None, None] * 16
         + (lanes >> 4) * 4 + np.arange(4)[None, None, None, None, :, None])
    k = (np.arange(nk64)[None, :, None, None, None, None] * 64 + np.arange(4)[None, None, None, :, None, None] * 16
         + (lanes & 15))
    got = ((bw[..., None] >> lanes.astype(np.uint64)) & np.uint64(1)).astype(bool)
    ok = (q < Lq) & (k < Lk)
    qq, kk = np.broadcast_arrays(np.minimum(q, Lq - 1), np.minimum(k, Lk - 1))
    want = keep[:, qq, kk]
    assert np.array_equal(got[:, ok], want[:, ok]), "backward layout"
    # per-lane layout of the one-workgroup-per-head forward: 32-bit word (bh, q64, k64, half, lane), element e = 8 qt + 4 tt + r
    # = keep(q = 64 q64 + 16 qt + (lane & 15), key = 64 k64 + 32 half + 16 tt + 4 (lane >> 4) + r) at bit (e >> 1) + 16 (e & 1)
    lw = bits[2 * half:].view(np.uint32).reshape(B * nh, nq16 // 4, nk64, 2, 64)
    e = np.arange(32)
    got = ((lw[..., None] >> ((e >> 1) + 16 * (e & 1)).astype(np.uint32)) & np.uint32(1)).astype(bool)   # (bh, q64, k64, half, lane, e)
    qt, tt, r = e >> 3, (e >> 2) & 1, e & 3
    q = np.arange(nq16 // 4)[:, None, None, None, None] * 64 + qt * 16 + (lanes & 15)[:, None]
    k = (np.arange(nk64)[None, :, None, None, None] * 64 + np.arange(2)[None, None, :, None, None] * 32 + tt * 16
         + (lanes >> 4)[:, None] * 4 + r)
    q, k = np.broadcast_arrays(q, k)
    ok = (q < Lq) & (k < Lk)
    want = keep[:, np.minimum(q, Lq - 1), np.minimum(k, Lk - 1)]
    assert np.array_equal(got[:, ok], want[:, ok]), "per-lane layout"


def test_attention_packed_layouts(ops):
    """Packed QKV / KV operands (the layouts the model uses) give the same result as separate tensors."""
    B, L, H, nh = 2, 77, 768, 12
    torch.manual_seed(6)
    qkv = torch.randn(B, L, 3 * H, device=DEV).to(torch.bfloat16).requires_grad_(True)
    o1 = ops.attention_self(qkv, None, None, nh)
    q, k, v = (qkv.detach()[..., i * H:(i + 1) * H].contiguous().requires_grad_(True) for i in range(3))
    o2 = ops.attention(q, k, v, None, None, nh)
    assert torch.equal(o1, o2)
    do = torch.randn_like(o1)
    o1.backward(do)
    o2.backward(do)
    assert torch.equal(qkv.grad[..., :H], q.grad) and torch.equal(qkv.grad[..., 2 * H:], v.grad)


def test_attention_full_size_softmax_properties(ops):
    """BASELINE.json full size (B=64, BEV self-attention): O rows are convex combinations of V rows
    (V = const -> O = const), and lse matches an fp32 recomputation on a slice."""
    B, L, nh, H = 64, 441, 12, 768
    torch.manual_seed(7)
    q = torch.randn(B, L, H, device=DEV).to(torch.bfloat16)
    k = torch.randn(B, L, H, device=DEV).to(torch.bfloat16)
    v = torch.full((B, L, H), 0.5, device=DEV).to(torch.bfloat16)
    o = ops.attention(q, k, v, None, None, nh)
    assert float((o.float() - 0.5).abs().max()) < 4e-3
    v = torch.randn(B, L, H, device=DEV).to(torch.bfloat16)
    o = ops.attention(q, k, v, None, None, nh)
    ref = _attn_ref(q[:2].float(), k[:2].float(), v[:2].float(), None, None, nh)
    assert float((o[:2].float() - ref).abs().max()) < 1e-2 * float(ref.abs().max())


# ----------------------------------------------------------------------------- K7 optimiser
def test_adamw_arena_matches_oracle_and_golden(ops):
    from vln_bevbert_amd.arena import ParamArena
    gld = load_golden("adamw")

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.from_numpy(gld["p0"]).clone())            # decayed (no "bias" in name)
            self.bias = torch.nn.Parameter(torch.from_numpy(gld["p0"]).clone())         # not decayed
            self.never = torch.nn.Parameter(torch.ones(5))                              # never receives a gradient

    m = M()
    arena = ParamArena(m, DEV, torch.bfloat16)
    for k in range(4):
        arena.zero_grad()
        g = torch.from_numpy(gld["grads"][k]).to(DEV)
        m.w.main_grad.add_(g)
        m.bias.main_grad.add_(g)
        arena.touch(m.w)
        arena.touch(m.bias)
        arena.clip_and_step(5e-5 * (k + 1) / 4, (0.9, 0.98), 1e-6, 0.01, max_norm=None)
        assert np.allclose(m.w.detach().cpu().numpy(), gld["wd0.01"][k], rtol=2e-6, atol=1e-7)
        assert np.allclose(m.bias.detach().cpu().numpy(), gld["wd0.0"][k], rtol=2e-6, atol=1e-7)
        assert torch.equal(m.w.compute.float(), m.w.detach().to(torch.bfloat16).float())    # shadow refreshed in-pass
    assert torch.equal(m.never.detach().cpu(), torch.ones(5))                           # .grad None => untouched


def test_grad_clip_coefficient(ops):
    from vln_bevbert_amd.arena import ParamArena
    m = torch.nn.Linear(300, 7)
    arena = ParamArena(m, DEV, torch.float32)
    g = torch.randn_like(arena.grads)
    arena.grads.copy_(g)
    for p in m.parameters():
        arena.touch(p)
    arena.clip_and_step(0.0, max_norm=5.0, grad_pre_scale=0.5)
    norm = float((0.5 * g).double().norm())
    assert abs(float(arena.grad_norm()) - norm) < 1e-3 * norm
    coef = float(arena._scalars[1])
    assert abs(coef - 0.5 * min(1.0, 5.0 / (norm + 1e-6))) < 1e-6


# ----------------------------------------------------------------------------- library GEMM through the C ABI
GEMM_CASES = [  # (rows, out_features, in_features)
    (5120, 768, 768), (28224, 2304, 768), (28224, 768, 3072), (3, 1, 768), (130, 40, 768), (2352, 768, 2048),
]


@pytest.mark.parametrize("M,N,K", GEMM_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lt_gemm_linear_fwd_dgrad_wgrad(ops, M, N, K, dtype):
    """bevbert_gemm (direct hipBLASLt) against fp64 matmuls for the three GEMMs of a Linear layer."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(DEV, dtype)
    w = (torch.randn(N, K, generator=g) * 0.05).to(DEV, dtype)
    b = torch.randn(N, generator=g).to(DEV, dtype)
    dy = torch.randn(M, N, generator=g).to(DEV, dtype)
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    fallbacks_before = sum(ops.GEMM_FALLBACKS.values())
    ops._linear_fwd(x, w, b)          # first launch of the problem: candidates are timed, the plan is final afterwards
    y = ops._linear_fwd(x, w, b)
    ref = x.double() @ w.double().t() + b.double()
    assert y.dtype == dtype and rel_err(y, ref) < tol
    dx = ops._linear_dgrad(dy, w)
    assert rel_err(dx, dy.double() @ w.double()) < tol
    dw = ops._linear_wgrad(dy, x)
    assert rel_err(dw, dy.double().t() @ x.double()) < tol
    if not ops._LT_UNSUPPORTED:      # the C-ABI path ran, not torch (plans may exist already when other tests ran first)
        assert sum(ops.GEMM_FALLBACKS.values()) == fallbacks_before, ops.GEMM_FALLBACKS
    # cached plan: same answer on the second call, and a strided (row-sliced) input is honoured
    assert torch.equal(ops._linear_fwd(x, w, b), y)
    wide = torch.randn(M, K + 64, generator=g).to(DEV, dtype)
    ys = ops._linear_fwd(wide[:, :K], w, None)
    assert rel_err(ys, wide[:, :K].double() @ w.double().t()) < tol


def test_lt_gemm_split_k_wgrad_into_sink(ops):
    """Strided-batch split-K partials + fused accumulate == one big dW GEMM (bf16 operands, fp32 sink)."""
    g = torch.Generator().manual_seed(7)
    M, N, K = 28224, 768, 768
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    dy = (torch.randn(M, N, generator=g) * 0.1).to(DEV, torch.bfloat16)
    sink = torch.ones(N, K, device=DEV)
    assert ops._split_k(M, N, K) > 1
    ops._wgrad_into(sink, dy, x)
    ops.WgradStream.flush_all()          # the fold into the sink is batched with the step's other pending reductions
    ref = 1.0 + dy.double().t() @ x.double()
    assert rel_err(sink, ref) < 1e-2
    sink32 = torch.ones(N, K, device=DEV)
    ops._wgrad_into(sink32, dy.float(), x.float())
    ops._wgrad_into(sink32, dy.float(), x.float())      # the same sink twice in one pass: successive launches, no race
    ops.WgradStream.flush_all()
    torch.cuda.synchronize()
    assert rel_err(sink32, 1.0 + 2 * (dy.float().double().t() @ x.float().double())) < 1e-4


@pytest.mark.parametrize("in_dtype,out_dtype", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                                (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("with_res", [False, True])
def test_dropout_add_matches_exported_mask(ops, in_dtype, out_dtype, with_res):
    """y = residual + dropout(x) against the library's own keep-mask hook (same (seed, offset) stream), fwd + bwd."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(37, 5, 768, generator=g).to(DEV, in_dtype).requires_grad_(True)
    res = torch.randn(37, 5, 768, generator=g).to(DEV, out_dtype).requires_grad_(True) if with_res else None
    p = 0.1
    ops.RT.new_step(1234)
    ops.RT.offset = 4096
    y = ops.dropout(x, p, True, residual=res, out_dtype=out_dtype)
    keep = ops.dropout_keep_mask(x.numel(), p, ops.RT.seed, 4096, DEV).view_as(x).float()
    assert 0.88 < float(keep.mean()) < 0.92
    ref = x.detach().float() * keep / (1 - p)
    if with_res:
        ref = ref + res.detach().float()
    tol = 1e-6 if out_dtype == torch.float32 else 1e-2
    assert y.dtype == out_dtype and rel_err(y, ref) < tol
    dy = torch.randn(y.shape, generator=g).to(DEV, out_dtype)
    grads = torch.autograd.grad(y, (x, res) if with_res else (x,), dy)
    assert grads[0].dtype == in_dtype and rel_err(grads[0], dy.float() * keep / (1 - p)) < tol
    if with_res:
        assert torch.equal(grads[1], dy)
    # eval mode / p == 0: identity (+ cast, + residual)
    y0 = ops.dropout(x, p, False, residual=res, out_dtype=out_dtype)
    ref0 = x.detach().to(out_dtype) if not with_res else res.detach() + x.detach().to(out_dtype)
    assert torch.equal(y0.detach(), ref0)


def test_splat_reads_store_rows_in_place(ops):
    """sample_rows: splatting straight out of a (N,P,C) fp16 store == splatting the gathered batch (bit-exact)."""
    g = torch.Generator().manual_seed(3)
    N, B, P, C, dim = 9, 5, 2352, 768, 21
    store = torch.randn(N, P, C, generator=g).to(DEV, torch.float16)
    sem_store = torch.randint(0, 40, (N, P), generator=g).to(DEV, torch.uint8)
    rows = torch.tensor([7, 0, 3, 7, 8], dtype=torch.int32, device=DEV)
    pts = ((torch.rand(B, P, 3, generator=g) - 0.5) * torch.tensor([12.0, 2.0, 12.0])).to(DEV)
    drop = (torch.rand(B, P, generator=g) < 0.05).to(DEV)
    cell, order, start = ops.bev_bin_points(pts, drop, dim, 0.5)
    for out_dtype in (torch.float32, torch.bfloat16):
        a, a_sem, a_mask = ops.bev_splat_mean(store, order, start, dim * dim, out_dtype=out_dtype, sems=sem_store, rows=rows)
        gathered = store.index_select(0, rows.long())
        b, b_sem, b_mask = ops.bev_splat_mean(gathered, order, start, dim * dim, out_dtype=out_dtype,
                                              sems=sem_store.index_select(0, rows.long()))
        assert torch.equal(a, b) and torch.equal(a_sem, b_sem) and torch.equal(a_mask, b_mask)
    assert a.shape == (B, dim * dim, C) and float(a.float().abs().sum()) > 0


@pytest.mark.parametrize("dtype,V,hw", [(torch.float32, 12, 14), (torch.float16, 12, 14), (torch.float16, 3, 3),
                                        (torch.uint8, 3, 3), (torch.float32, 1, 1)])
def test_gather_views_of_selected_nodes(ops, dtype, V, hw):
    """bevbert_gm_gather_views == index_select + zeroed padding slots (agent.py:150-156), for every word width the launch
    picks (16 / 4 / 2 / 1 bytes by the row size and the alignment), rows repeated, slots dead."""
    from vln_bevbert_amd import lib
    g = torch.Generator().manual_seed(5)
    N, n_out = 11, 23
    store = (torch.rand(N + 1, V, hw, hw, generator=g) * 200).to(DEV).to(dtype)
    rows = torch.randint(0, N, (n_out,), generator=g).to(DEV, torch.int32)
    live = (torch.rand(n_out, generator=g) < 0.7).to(DEV)
    for view in (store[:N], store[1:]):                       # the second start is aligned to the element size only
        out = torch.full((n_out, V, hw, hw), 7, dtype=dtype, device=DEV)
        lib.call("bevbert_gm_gather_views", view.data_ptr(), rows.data_ptr(), live.data_ptr(), out.data_ptr(), n_out,
                 V * hw * hw * view.element_size(), lib.stream())
        ref = view.index_select(0, rows.long()) * live.to(dtype)[:, None, None, None]
        assert torch.equal(out, ref)
    lib.call("bevbert_gm_gather_views", store.data_ptr(), rows.data_ptr(), live.data_ptr(), out.data_ptr(), 0, 4, lib.stream())


def test_lt_gemm_tuning_table_round_trip(ops, tmp_path):
    """The choices of autotuned plans can be exported and re-imported (bevbert_gemm_tuning_export / _import)."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1536, 768, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(1280, 768, generator=g) * 0.05).to(DEV, torch.bfloat16)
    ops._linear_fwd(x, w, None)                           # first launch: times the candidates of this problem
    y = ops._linear_fwd(x, w, None)
    path = str(tmp_path / "tuning.txt")
    rows = ops.save_gemm_tuning_table(path)
    text = open(path).read()
    assert rows >= 1 and text.startswith("# bevbert gemm tuning v1 hipblaslt ")
    assert any(line.startswith("1536.1280.768.0.1.") for line in text.splitlines())
    assert ops.load_gemm_tuning_table(path) == rows
    assert torch.equal(ops._linear_fwd(x, w, None), y)
    from vln_bevbert_amd import lib
    assert lib.load().bevbert_gemm_tuning_import(b"# some other library\nfoo 1 2\n") == -3
