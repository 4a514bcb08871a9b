"""Row-wise hand-written kernels and their autograd Functions (ops.py re-exports everything here): bias + dropout + residual +
LayerNorm (bf16 and fp32 residual streams), dropout, bias + GELU, the SAP / MLM loss tails, embedding sums, the global-map
segment gather and the BEV lift / splat."""
import math

import torch

from . import lib
from .lib import dtype_code, ptr, stream
from .ops_core import RT, _compute, _f32, _mark_touched, _sink, call
from .ops_reduce import ReduceQueue, WgradStream, _on_launch_stream, _partial_rows


def embedding_grad_small(ids, d, sink, table_rows):
    """sink (fp32 arena view, table_rows x H) += scatter-sum of d's rows by ids, for tables of a few rows: sliced partial
    sums in the scratch ring (bevbert_embedding_grad_sliced), folded in by the step's batched column reduction
    (ReduceQueue / bevbert_multi_finalize: 16 row lanes per 64 columns, so hundreds of slices are fine)."""
    rows, H = d.shape
    per = 64
    while (rows + per - 1) // per * table_rows > 4096:        # keep the launch at a few thousand workgroups
        per *= 2
    slices = (rows + per - 1) // per
    part = RT.scratch.alloc(slices * table_rows * H * 4, d.device)
    call("bevbert_embedding_grad_sliced", ptr(ids), ptr(d), part, rows, H, table_rows, per, dtype_code(d), stream())
    ReduceQueue.add(part, slices, 1, table_rows * H, (sink.data_ptr(), None, None))


# ----------------------------------------------------------------------------- K3 LayerNorm family
class _BiasDropResLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, residual, gamma, beta, eps, drop_p, inplace_z, post1=None, post2=None, return_z=False):
        assert x.is_contiguous() and x.dim() >= 2
        H = x.shape[-1]
        rows = x.numel() // H
        y = torch.empty_like(x)
        need_grad = any(ctx.needs_input_grad)
        plain = bias is None and residual is None and drop_p == 0
        z = (x if (inplace_z or plain) else torch.empty_like(x)) if need_grad else None
        mean = torch.empty(rows, dtype=torch.float32, device=x.device) if need_grad else None
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if need_grad else None
        off = RT.next_offset(x.numel()) if drop_p > 0 else 0
        if residual is not None:
            assert residual.is_contiguous() and residual.shape == x.shape and residual.dtype == x.dtype
        if post1 is not None or post2 is not None:
            # y = LN(x + bias) + post1 + post2: the sums that follow the LayerNorm ride on its store (rowops.hip)
            assert residual is None and drop_p == 0, "post terms: plain bias + LayerNorm only"
            for t in (post1, post2):
                assert t is None or (t.is_contiguous() and t.shape == x.shape and t.dtype == x.dtype)
            call("bevbert_layernorm_post_fwd", ptr(x), ptr(_f32(bias)) if bias is not None else None, ptr(_f32(gamma)),
                 ptr(_f32(beta)), ptr(post1), ptr(post2), ptr(y), None if plain else ptr(z), ptr(mean), ptr(rstd), rows,
                 H, float(eps), dtype_code(x), stream())
        else:
            call("bevbert_bias_dropout_residual_layernorm_fwd", ptr(x), ptr(_f32(bias)) if bias is not None else None,
                 ptr(residual), ptr(_f32(gamma)), ptr(_f32(beta)), ptr(y), None if plain else ptr(z), ptr(mean),
                 ptr(rstd), rows, H, float(eps), dtype_code(x), float(drop_p), RT.seed, off, stream())
        ctx.save_for_backward(z, mean, rstd)
        ctx.params = (bias, gamma, beta)
        ctx.cfg = (rows, H, float(drop_p), RT.seed, off, residual is not None)
        ctx.posts = (post1 is not None, post2 is not None)
        ctx.return_z = return_z
        ctx.set_materialize_grads(False)       # (y, z): the unused one arrives as None, not as a dense zero tensor
        if return_z:
            # pre-norm blocks (transformer.py:170-182): z = residual + dropout(x + bias) is the NEW residual stream and
            # y = LayerNorm(z) feeds the next sub-layer; the gradient arriving at z is added to LayerNorm's input gradient
            # inside the backward kernel (bevbert_layernorm_bwd_add)
            assert need_grad and z is not None and z is not x
            return y, z.view_as(z)
        return y

    @staticmethod
    def backward(ctx, dy, dz_in=None):
        z, mean, rstd = ctx.saved_tensors
        bias, gamma, beta = ctx.params
        rows, H, drop_p, seed, off, has_res = ctx.cfg
        if dy is None:              # only z was used downstream: LayerNorm itself contributes nothing
            dy = torch.zeros_like(z)
        dy = dy.contiguous()
        add = None
        if ctx.return_z and dz_in is not None:
            add = dz_in.contiguous()
            assert add.dtype == dy.dtype and add.shape == dy.shape
        dz = torch.empty_like(dy)
        dx = torch.empty_like(dy) if (drop_p > 0 and has_res) else None
        dev = dy.device
        ws = RT.workspace(dev, lib.load().bevbert_colsum_workspace_floats(3 * H))
        outs = []
        for p in (gamma, beta, bias):
            if p is None:
                outs.append((None, None, 0))
            elif p is bias and not getattr(p, "requires_grad", True):
                outs.append((None, None, None))          # a frozen bias (e.g. fix_lang_embedding): no gradient wanted
            elif _sink(p) is not None:
                outs.append((_sink(p), None, 1))
                _mark_touched(p)
            else:
                t = torch.empty(H, dtype=torch.float32, device=dev)
                outs.append((t, t, 0))
        (dg, rg, ag), (db, rb, ab), (dbi, rbi, abi) = outs
        assert ag == ab and (bias is None or abi is None or abi == ag), "mixed arena / plain parameters in one LayerNorm"
        # without a residual branch only dx is needed (it is the single input gradient)
        if not has_res and drop_p > 0:
            dx, dz_ptr = dz, None
        else:
            dz_ptr = dz
        if ag == 1 and WgradStream.DEFER_FINALIZE and dev.type == "cuda":
            # arena parameters: the kernel leaves its per-block partial sums in the scratch ring; the second stage of
            # the reduction joins the step's other pending reductions (ReduceQueue: one launch, off the critical path)
            nb = _partial_rows(rows)
            part = RT.scratch.alloc(nb * 3 * H * 4, dev)
            call("bevbert_layernorm_bwd_add", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(_f32(gamma)), ptr(dz_ptr), ptr(dx),
                 ptr(add), None, None, None, part, rows, H, dtype_code(dy), drop_p, seed, off, 1, stream())
            ReduceQueue.add(part, nb, 3, H, (ptr(dg), ptr(db), ptr(dbi)))
        else:
            call("bevbert_layernorm_bwd_add", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(_f32(gamma)), ptr(dz_ptr), ptr(dx),
                 ptr(add), ptr(dg), ptr(db), ptr(dbi), ptr(ws), rows, H, dtype_code(dy), drop_p, seed, off, ag, stream())
        gx = dx if dx is not None else dz
        gres = dz if has_res else None
        cast = lambda r, p: None if r is None else r.to(p.dtype)
        g1, g2 = (dy if has else None for has in ctx.posts)      # the post terms were added after the affine
        return (gx, cast(rbi, bias) if bias is not None else None, gres, cast(rg, gamma), cast(rb, beta), None, None, None,
                g1, g2, None)


class _BiasDropResLN32(torch.autograd.Function):
    """LayerNorm(dropout(x + bias) + residual) with the fp32 residual stream of ``RT.res32``: x bf16 (a GEMM output),
    residual fp32 (the previous block's ``y32``) or bf16 (where a stream starts); returns (y16, y32).  The backward sums the
    two output gradients in the kernel (bf16 from the GEMMs that read y16, fp32 from the residual add that read y32) and
    returns dz in fp32 to an fp32 residual."""

    @staticmethod
    def forward(ctx, x, bias, residual, gamma, beta, eps, drop_p):
        assert x.is_contiguous() and x.dtype == torch.bfloat16 and x.dim() >= 2
        H = x.shape[-1]
        rows = x.numel() // H
        need_grad = any(ctx.needs_input_grad)
        y16 = torch.empty_like(x)
        y32 = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        z32 = torch.empty(x.shape, dtype=torch.float32, device=x.device) if need_grad else None
        mean = torch.empty(rows, dtype=torch.float32, device=x.device) if need_grad else None
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if need_grad else None
        off = RT.next_offset(x.numel()) if drop_p > 0 else 0
        assert residual.is_contiguous() and residual.shape == x.shape and residual.dtype in (torch.float32, torch.bfloat16)
        call("bevbert_layernorm_res32_fwd", ptr(x), ptr(_f32(bias)) if bias is not None else None, ptr(residual),
             dtype_code(residual), ptr(_f32(gamma)), ptr(_f32(beta)), ptr(y16), ptr(y32), ptr(z32), ptr(mean), ptr(rstd),
             rows, H, float(eps), float(drop_p), RT.seed, off, stream())
        ctx.save_for_backward(z32, mean, rstd)
        ctx.params = (bias, gamma, beta)
        ctx.cfg = (rows, H, float(drop_p), RT.seed, off, residual.dtype)
        ctx.set_materialize_grads(False)       # an unused output arrives as None (backward handles it), not as dense zeros
        return y16, y32

    @staticmethod
    def backward(ctx, dy16, dy32):
        z32, mean, rstd = ctx.saved_tensors
        bias, gamma, beta = ctx.params
        rows, H, drop_p, seed, off, res_dtype = ctx.cfg
        dev = z32.device
        if dy16 is None and dy32 is None:
            dy32 = torch.zeros_like(z32)
        dy16 = dy16.contiguous() if dy16 is not None else None
        dy32 = dy32.contiguous() if dy32 is not None else None
        dz32 = torch.empty(z32.shape, dtype=res_dtype, device=dev)     # the residual's gradient, in the residual's dtype
        dx16 = torch.empty(z32.shape, dtype=torch.bfloat16, device=dev)
        outs = []
        for p in (gamma, beta, bias):
            if p is None:
                outs.append((None, None, 0))
            elif p is bias and not getattr(p, "requires_grad", True):
                outs.append((None, None, None))
            elif _sink(p) is not None:
                outs.append((_sink(p), None, 1))
                _mark_touched(p)
            else:
                t = torch.empty(H, dtype=torch.float32, device=dev)
                outs.append((t, t, 0))
        (dg, rg, ag), (db, rb, ab), (dbi, rbi, abi) = outs
        assert ag == ab and (bias is None or abi is None or abi == ag), "mixed arena / plain parameters in one LayerNorm"
        if ag == 1 and WgradStream.DEFER_FINALIZE and dev.type == "cuda":
            nb = _partial_rows(rows)
            part = RT.scratch.alloc(nb * 3 * H * 4, dev)
            call("bevbert_layernorm_res32_bwd", ptr(dy16), ptr(dy32), ptr(z32), ptr(mean), ptr(rstd), ptr(_f32(gamma)),
                 ptr(dz32), ptr(dx16), None, None, None, part, rows, H, drop_p, seed, off, 1, dtype_code(dz32), stream())
            ReduceQueue.add(part, nb, 3, H, (ptr(dg), ptr(db), ptr(dbi)))
        else:
            ws = RT.workspace(dev, lib.load().bevbert_colsum_workspace_floats(3 * H))
            call("bevbert_layernorm_res32_bwd", ptr(dy16), ptr(dy32), ptr(z32), ptr(mean), ptr(rstd), ptr(_f32(gamma)),
                 ptr(dz32), ptr(dx16), ptr(dg), ptr(db), ptr(dbi), ptr(ws), rows, H, drop_p, seed, off, ag, dtype_code(dz32),
                 stream())
        cast = lambda r, p: None if r is None else r.to(p.dtype)
        gres = dz32
        return dx16, cast(rbi, bias) if bias is not None else None, gres, cast(rg, gamma), cast(rb, beta), None, None


def bias_dropout_residual_layernorm(x, bias, residual, gamma, beta, eps, drop_p=0.0, training=False,
                                    inplace_z=True):
    """LayerNorm(dropout(x + bias) + residual)  -- vilmodel.py:150-154,189-193."""
    p = float(drop_p) if training else 0.0
    if RT.res32 and residual is not None and x.dtype == torch.bfloat16 and x.is_cuda:
        # fp32 residual stream: the previous block left its fp32 output on the bf16 tensor the model passes around
        r32 = getattr(residual, "_res32", None)
        y16, y32 = _BiasDropResLN32.apply(x, bias, r32 if r32 is not None else residual, gamma, beta, eps, p)
        y16._res32 = y32
        return y16
    return _BiasDropResLN.apply(x, bias, residual, gamma, beta, eps, p, inplace_z, None, None)


def bias_dropout_residual_prenorm(x, bias, residual, gamma, beta, eps, drop_p=0.0, training=False):
    """(LayerNorm(z), z) with z = residual + dropout(x + bias) (residual None: z = dropout(x + bias), the dropout in front of
    a pre-norm stack): one launch for the residual add of a pre-norm block AND the
    LayerNorm that opens the next sub-layer (transformer.py:170-182); backward likewise (the gradient reaching z from the
    rest of the stream is folded into the LayerNorm backward kernel).  Inference / no-grad callers get the two tensors
    from the same launch too."""
    p = float(drop_p) if training else 0.0
    if not (torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad))):
        assert x.is_contiguous() and (residual is None or (residual.is_contiguous() and residual.shape == x.shape and residual.dtype == x.dtype))
        H = x.shape[-1]
        rows = x.numel() // H
        y, z = torch.empty_like(x), torch.empty_like(x)
        off = RT.next_offset(x.numel()) if p > 0 else 0
        call("bevbert_bias_dropout_residual_layernorm_fwd", ptr(x), ptr(_f32(bias)) if bias is not None else None,
             ptr(residual), ptr(_f32(gamma)), ptr(_f32(beta)), ptr(y), ptr(z), None, None, rows, H, float(eps),
             dtype_code(x), p, RT.seed, off, stream())
        return y, z
    return _BiasDropResLN.apply(x, bias, residual, gamma, beta, eps, p, False, None, None, True)


def bias_layernorm_plus(x, bias, gamma, beta, eps, post1, post2=None):
    """(LayerNorm(x + bias) + post1) + post2 in one launch -- the sums of the embedding compositions
    (vilmodel.py:494-532, 589-593); fp32 results equal the separate adds bit for bit (same order of additions)."""
    return _BiasDropResLN.apply(x, bias, None, gamma, beta, eps, 0.0, True, post1.contiguous(),
                                None if post2 is None else post2.contiguous())


class _SmallKLinearLN(torch.autograd.Function):
    """y = (LayerNorm(feat W^T + b) + post1) + table[idx] with K = feat.shape[-1] <= 16 (smallk.hip): the projection is
    recomputed inside the LayerNorm kernels, forward and backward; parameter gradients go straight into the arena."""

    @staticmethod
    def forward(ctx, feat, weight, bias, gamma, beta, eps, post1, table, table_c, idx):
        K, H = feat.shape[-1], weight.shape[0]
        rows = feat.numel() // K
        assert feat.dtype == torch.float32 and feat.is_contiguous() and weight.shape == (H, K)
        assert post1.is_contiguous() and post1.shape[-1] == H and post1.numel() == rows * H
        need_grad = any(ctx.needs_input_grad)
        y = torch.empty_like(post1)
        mean = torch.empty(rows, dtype=torch.float32, device=feat.device) if need_grad else None
        rstd = torch.empty(rows, dtype=torch.float32, device=feat.device) if need_grad else None
        if idx is not None:
            idx = idx.contiguous()
            assert idx.dtype == torch.int64 and idx.numel() == rows and table_c.dtype == post1.dtype
        call("bevbert_smallk_linear_layernorm_fwd", ptr(feat), ptr(weight), ptr(bias), ptr(_f32(gamma)), ptr(_f32(beta)),
             ptr(post1), ptr(table_c) if idx is not None else None, ptr(idx), ptr(y), ptr(mean), ptr(rstd), rows, K, H,
             float(eps), dtype_code(y), stream())
        ctx.save_for_backward(feat, mean, rstd, idx)
        ctx.params = (weight, bias, gamma, beta, table)
        return y

    @staticmethod
    def backward(ctx, dy):
        feat, mean, rstd, idx = ctx.saved_tensors
        weight, bias, gamma, beta, table = ctx.params
        dy = dy.contiguous()
        K, H = feat.shape[-1], weight.shape[0]
        rows = feat.numel() // K
        sinks = []
        for p in (weight, bias, gamma, beta):
            s = _sink(p) if (p is not None and p.requires_grad) else None
            if s is not None:
                _mark_touched(p)
            sinks.append(s)
        ws = RT.scratch.alloc(int(lib.load().bevbert_smallk_workspace_floats(rows, K, H)) * 4, dy.device)
        call("bevbert_smallk_linear_layernorm_bwd", ptr(dy), ptr(feat), ptr(weight), ptr(bias), ptr(mean), ptr(rstd),
             ptr(_f32(gamma)), ptr(sinks[0]), ptr(sinks[1]), ptr(sinks[2]), ptr(sinks[3]), ws, rows, K, H, dtype_code(dy),
             stream())
        if idx is not None and table.requires_grad:
            _mark_touched(table)
            embedding_grad_small(idx.reshape(-1), dy.reshape(-1, H), _sink(table), table.shape[0])
        return None, None, None, None, None, None, dy, None, None, None


def smallk_linear_layernorm_plus_supported(feat, lin, ln, post1, emb=None):
    """The fused kernels take over when everything lives where they expect it: device tensors, fp32 master parameters in
    the gradient arena (their gradients are accumulated in place), a hidden width the row kernels are instantiated for."""
    H, K = lin.weight.shape
    ps = [lin.weight, lin.bias, ln.weight, ln.bias] + ([emb.weight] if emb is not None else [])
    return (feat.is_cuda and K <= 16 and H in (256, 512, 768, 1024) and (K + 8) * H * 4 + 512 <= 65536
            and WgradStream.DEFER_FINALIZE and all(p.dtype == torch.float32 and (not p.requires_grad or _sink(p) is not None) for p in ps)
            and post1.dtype in (torch.float32, torch.bfloat16))


def smallk_linear_layernorm_plus(feat, lin, ln, eps, post1, emb=None, idx=None):
    """(LN(lin(feat)) + post1) + emb(idx) -- vilmodel.py:507-518, 577-583, 589-593."""
    tc = None if emb is None else _compute(emb.weight)
    return _SmallKLinearLN.apply(feat.to(torch.float32).contiguous(), lin.weight, lin.bias, ln.weight, ln.bias, eps,
                                 post1.contiguous(), None if emb is None else emb.weight, tc, idx)


def layernorm(x, gamma, beta, eps):
    return _BiasDropResLN.apply(x.contiguous(), None, None, gamma, beta, eps, 0.0, False, None, None)


# ----------------------------------------------------------------------------- dropout (+ residual, + cast)
class _DropoutAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, p, out_dtype):
        assert x.is_contiguous() and x.numel() % 4 == 0
        out_dtype = out_dtype or x.dtype
        y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        if residual is not None:
            assert residual.is_contiguous() and residual.shape == x.shape and residual.dtype == out_dtype
        off = RT.next_offset(x.numel())
        call("bevbert_dropout_add", ptr(x), ptr(residual), ptr(y), x.numel(), dtype_code(x), dtype_code(y), p, RT.seed,
             off, stream())
        ctx.cfg = (p, RT.seed, off, x.dtype, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, off, in_dtype, has_res = ctx.cfg
        dx = None
        if ctx.needs_input_grad[0]:
            dy = dy.contiguous()
            dx = torch.empty_like(dy)
            call("bevbert_dropout_add", ptr(dy), None, ptr(dx), dy.numel(), dtype_code(dy), dtype_code(dx), p, seed, off,
                 stream())
            if dx.dtype != in_dtype:
                dx = dx.to(in_dtype)
        return dx, (dy if has_res else None), None, None


def dropout(x, p, training, residual=None, out_dtype=None):
    """residual + nn.Dropout(p)(x) on the library's counter-based mask stream (reproducible from (seed, step) alone);
    ``out_dtype`` fuses the cast of fp32 loader features to the compute dtype."""
    if not training or p <= 0.0:
        y = x if out_dtype is None or out_dtype == x.dtype else x.to(out_dtype)
        return y if residual is None else residual + y
    return _DropoutAdd.apply(x.contiguous(), residual, float(p), out_dtype)


# ----------------------------------------------------------------------------- SAP loss tail
class _SapLoss(torch.autograd.Function):
    """loss (B,) of forward_sap behind the three heads (pretrain_cmt.py:225-275) in one launch; see bevbert_sap_loss_fwd."""

    @staticmethod
    def forward(ctx, graw, lraw, fraw, visited, gmap_lens, nav_masks, cand_idxs, src, vis_c, glabels, llabels):
        B, G = graw.shape
        K = lraw.shape[1]
        dev = graw.device
        assert lraw.dtype == graw.dtype and (fraw is None or fraw.dtype == graw.dtype)
        graw, lraw = graw.contiguous(), lraw.contiguous()
        fraw = None if fraw is None else fraw.contiguous()
        buf = torch.empty(B * (G + K + 2), dtype=torch.float32, device=dev)
        loss, dG, dL, dF = buf[:B], buf[B:B + B * G], buf[B + B * G:B + B * (G + K)], buf[B + B * (G + K):]
        as_u8 = lambda t: t.contiguous().view(torch.uint8)
        call("bevbert_sap_loss_fwd", ptr(graw), ptr(lraw), ptr(fraw), ptr(as_u8(visited)), ptr(gmap_lens.contiguous()),
             ptr(as_u8(nav_masks)), ptr(cand_idxs.contiguous()), ptr(src.contiguous()), ptr(as_u8(vis_c)),
             ptr(glabels.contiguous()), ptr(llabels.contiguous()), ptr(loss), ptr(dG), ptr(dL), ptr(dF), B, G, K,
             nav_masks.shape[1], dtype_code(graw), stream())
        ctx.save_for_backward(buf)
        ctx.dims = (B, G, K, graw.dtype, fraw is not None)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (buf,) = ctx.saved_tensors
        B, G, K, dt, has_f = ctx.dims
        dG, dL, dF = buf[B:B + B * G], buf[B + B * G:B + B * (G + K)], buf[B + B * (G + K):]
        out = torch.empty(B * (G + K + 1), dtype=dt, device=buf.device)
        dgr, dlr, dfr = out[:B * G].view(B, G), out[B * G:B * (G + K)].view(B, K), out[B * (G + K):].view(B, 1)
        call("bevbert_sap_loss_bwd", ptr(dG), ptr(dL), ptr(dF), ptr(dloss.contiguous().float()), ptr(dgr), ptr(dlr),
             ptr(dfr) if has_f else None, B, G, K, dtype_code(out), stream())
        return (dgr, dlr, dfr if has_f else None) + (None,) * 8


def sap_loss_supported(graw, lraw):
    return graw.is_cuda and graw.shape[1] <= 64 and lraw.shape[1] <= 62 and graw.dtype in (torch.float32, torch.bfloat16)


def sap_loss(graw, lraw, fraw, visited, gmap_lens, nav_masks, cand_idxs, src, vis_c, glabels, llabels):
    """(B,) loss of the SAP task from the raw head outputs: graw (B,G), lraw (B,K), fraw (B,1) or None."""
    return _SapLoss.apply(graw, lraw, fraw, visited, gmap_lens, nav_masks, cand_idxs, src, vis_c, glabels, llabels)


class _CrossEntropy(torch.autograd.Function):
    """F.cross_entropy(logits.float(), target, reduction="none") without the fp32 copy of the logits."""

    @staticmethod
    def forward(ctx, logits, target):
        rows, C = logits.shape
        logits = logits.contiguous()
        out = torch.empty(2, rows, dtype=torch.float32, device=logits.device)
        call("bevbert_cross_entropy_fwd", ptr(logits), ptr(target.contiguous()), ptr(out[0]), ptr(out[1]), rows, C,
             dtype_code(logits), stream())
        ctx.save_for_backward(logits, target, out)
        return out[0]

    @staticmethod
    def backward(ctx, dloss):
        logits, target, out = ctx.saved_tensors
        rows, C = logits.shape
        d = torch.empty_like(logits)
        call("bevbert_cross_entropy_bwd", ptr(logits), ptr(target.contiguous()), ptr(out[1]),
             ptr(dloss.contiguous().float()), ptr(d), rows, C, dtype_code(logits), stream())
        return d, None


def cross_entropy_rows(logits, target):
    """(rows,) fp32 losses of (rows, C) logits in the compute dtype (the MLM head's vocabulary rows)."""
    if logits.is_cuda and logits.dtype in (torch.float32, torch.bfloat16):
        return _CrossEntropy.apply(logits, target)
    return torch.nn.functional.cross_entropy(logits.float(), target, reduction="none")


# ----------------------------------------------------------------------------- K4 bias + GELU
class _BiasGelu(torch.autograd.Function):
    """act(x + bias); ``act``: "gelu" (erf) or "relu" -- the C ABI has one entry pair per activation."""

    @staticmethod
    def forward(ctx, x, bias, act="gelu"):
        assert x.is_contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        y = torch.empty_like(x)
        call(f"bevbert_bias_{act}_fwd", ptr(x), ptr(_f32(bias)), ptr(y), rows, C, dtype_code(x), stream())
        ctx.save_for_backward(x)
        ctx.bias = bias
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        bias = ctx.bias
        entry = f"bevbert_bias_{ctx.act}_bwd"
        C = x.shape[-1]
        rows = x.numel() // C
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        ws = RT.workspace(dy.device, lib.load().bevbert_colsum_workspace_floats(C))
        sink = _sink(bias)
        if sink is not None:
            _mark_touched(bias)
            if WgradStream.DEFER_FINALIZE:      # second reduction stage batched with the step's others (see _BiasDropResLN)
                nb = _partial_rows(rows)
                part = RT.scratch.alloc(nb * C * 4, dy.device)
                call(entry, ptr(dy), ptr(x), ptr(_f32(bias)), ptr(dx), None, part, rows, C, dtype_code(dy), 1, stream())
                ReduceQueue.add(part, nb, 1, C, (ptr(sink), None, None))
                return dx, None, None
            call(entry, ptr(dy), ptr(x), ptr(_f32(bias)), ptr(dx), ptr(sink), ptr(ws), rows, C, dtype_code(dy), 1, stream())
            return dx, None, None
        db = torch.empty(C, dtype=torch.float32, device=dy.device)
        call(entry, ptr(dy), ptr(x), ptr(_f32(bias)), ptr(dx), ptr(db), ptr(ws), rows, C, dtype_code(dy), 0, stream())
        return dx, db.to(bias.dtype), None


class _WeightedMean(torch.autograd.Function):
    """sum_i w[i] x[i] / denom as ONE launch each way (loss.mean() of the step, train_r2r.py:263, with the zero-weight
    padding rows of a static batch): x fp32 (n), w fp32 (n) or None, denom a device scalar or a Python number."""

    @staticmethod
    def forward(ctx, x, w, denom):
        x = x.contiguous()
        assert x.dtype == torch.float32 and x.dim() == 1 and (w is None or (w.dtype == torch.float32 and w.shape == x.shape))
        out = torch.empty((), dtype=torch.float32, device=x.device)
        dd = denom if torch.is_tensor(denom) else None
        call("bevbert_weighted_mean_fwd", ptr(x), ptr(w), ptr(dd), 1.0 if dd is not None else float(denom), x.numel(), ptr(out),
             stream())
        ctx.w, ctx.denom, ctx.n = w, denom, x.numel()
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous().to(torch.float32)
        dx = torch.empty(ctx.n, dtype=torch.float32, device=dout.device)
        dd = ctx.denom if torch.is_tensor(ctx.denom) else None
        call("bevbert_weighted_mean_bwd", ptr(dout), ptr(ctx.w), ptr(dd), 1.0 if dd is not None else float(ctx.denom), ctx.n,
             ptr(dx), stream())
        return dx, None, None


def weighted_mean(x, w=None, denom=None):
    """(w * x).sum() / denom; denom defaults to the number of elements (plain mean)."""
    if not x.is_cuda:
        s = (x if w is None else x * w).sum()
        return s / (x.numel() if denom is None else denom)
    return _WeightedMean.apply(x, w, x.numel() if denom is None else denom)


class _BceRows(torch.autograd.Function):
    """per_row[r] = sum_c binary_cross_entropy_with_logits(logits[r, c], labels[idx[r], c]) -- the semantic head's loss
    (pretrain_cmt.py:391-441) with the uint8 label rows read through the selection index, one launch each way."""

    @staticmethod
    def forward(ctx, logits, labels, idx):
        logits = logits.contiguous()
        rows, C = logits.shape
        assert labels.dtype == torch.uint8 and labels.is_contiguous() and labels.shape[-1] == C
        out = torch.empty(rows, dtype=torch.float32, device=logits.device)
        call("bevbert_bce_rows_fwd", ptr(logits), ptr(labels), ptr(idx), ptr(out), rows, C, dtype_code(logits), stream())
        ctx.save_for_backward(logits, labels, idx)
        return out

    @staticmethod
    def backward(ctx, g):
        logits, labels, idx = ctx.saved_tensors
        rows, C = logits.shape
        dx = torch.empty_like(logits)
        call("bevbert_bce_rows_bwd", ptr(logits), ptr(labels), ptr(idx), ptr(g.contiguous().float()), ptr(dx), rows, C,
             dtype_code(logits), stream())
        return dx, None, None


def bce_rows(logits, labels, idx=None):
    return _BceRows.apply(logits, labels, idx)


@torch.no_grad()
def sem_select(mask1, mask2, cap, classes):
    """(idx int64 (cap), valid fp32 (cap), denom fp32 scalar) of the cells with mask1 & mask2 set (bool / uint8 tensors);
    see include/bevbert_hip.h bevbert_sem_select."""
    m1 = mask1.reshape(-1).contiguous()
    m2 = None if mask2 is None else mask2.reshape(-1).contiguous()
    assert m1.dtype in (torch.bool, torch.uint8) and (m2 is None or (m2.dtype in (torch.bool, torch.uint8) and m2.numel() == m1.numel()))
    dev = m1.device
    idx = torch.empty(cap, dtype=torch.int64, device=dev)
    valid = torch.empty(cap, dtype=torch.float32, device=dev)
    denom = torch.empty((), dtype=torch.float32, device=dev)
    call("bevbert_sem_select", ptr(m1), ptr(m2), m1.numel(), cap, classes, ptr(idx), ptr(valid), ptr(denom), stream())
    return idx, valid, denom


class _TakeRows(torch.autograd.Function):
    """rows = x2d[idx] for one or two index vectors of ONE tensor (a second selection from the same activations -- the
    centre cell next to the candidate cells of the SAP head -- shares the backward's zero-initialised gradient tensor:
    autograd would otherwise materialise two dense gradients and add them)."""

    @staticmethod
    def forward(ctx, x, idx, idx2=None):
        assert x.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous()
        H = x.shape[-1]
        outs = []
        for ix in (idx, idx2):
            if ix is None:
                continue
            o = torch.empty((ix.numel(), H), dtype=x.dtype, device=x.device)
            call("bevbert_rows_gather", ptr(x), ptr(ix), ptr(o), ix.numel(), H, dtype_code(x), stream())
            outs.append(o)
        ctx.save_for_backward(idx, idx2)
        ctx.shape = x.shape
        ctx.set_materialize_grads(False)       # an unused selection arrives as None, not as a dense zero tensor
        return tuple(outs) if idx2 is not None else outs[0]

    @staticmethod
    def backward(ctx, *grads):
        idx, idx2 = ctx.saved_tensors
        some = next((g for g in grads if g is not None), None)
        if some is None:
            return None, None, None
        H = ctx.shape[-1]
        dx = torch.empty(ctx.shape, dtype=some.dtype, device=some.device)
        call("bevbert_zero", ptr(dx), dx.numel() * dx.element_size(), stream())
        first = True
        for ix, g in zip((idx, idx2), grads):
            if ix is None or g is None:
                continue
            g = g.contiguous()
            call("bevbert_rows_scatter", ptr(ix), ptr(g), ptr(dx), ix.numel(), H, dtype_code(g), 0 if first else 1, stream())
            first = False
        return dx, None, None


def take_rows(x, idx, idx2=None):
    """x.reshape(-1, H)[idx] (and [idx2]) -- pretrain_cmt.py:254-256,321-326,403-410; the backward is a memset and a
    deterministic scatter (duplicate indices are summed in row order, no atomics)."""
    return _TakeRows.apply(x.contiguous(), idx, idx2)


def bias_gelu(x, bias):
    """gelu_erf(x + bias) -- vilmodel.py:31-37,177-180."""
    return _BiasGelu.apply(x, bias)


def bias_relu(x, bias):
    """relu(x + bias): the prediction heads' Linear -> ReLU (pretrain_cmt.py:34-71), the Linear's bias on the activation."""
    return _BiasGelu.apply(x, bias, "relu")


# ----------------------------------------------------------------------------- K5 embeddings
class _EmbedLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, word, pos, typ, gamma, beta, eps, type_index, word_c, pos_c, typ_c, pad_idx=-1):
        B, L = ids.shape
        H = word_c.shape[1]
        rows = B * L
        y = torch.empty(B, L, H, dtype=word_c.dtype, device=ids.device)
        need_grad = any(ctx.needs_input_grad)
        z = torch.empty_like(y) if need_grad else None
        mean = torch.empty(rows, dtype=torch.float32, device=ids.device) if need_grad else None
        rstd = torch.empty(rows, dtype=torch.float32, device=ids.device) if need_grad else None
        ids = ids.contiguous()
        call("bevbert_embed_sum_layernorm_fwd", ptr(ids), ptr(word_c), ptr(pos_c), ptr(typ_c[type_index]),
             ptr(_f32(gamma)), ptr(_f32(beta)), ptr(y), ptr(z), ptr(mean), ptr(rstd), rows, L, H, float(eps),
             dtype_code(y), 0.0, 0, 0, stream())
        ctx.save_for_backward(ids, z, mean, rstd)
        ctx.params = (word, pos, typ, gamma, beta, type_index, int(pad_idx))
        return y

    @staticmethod
    def backward(ctx, dy):
        ids, z, mean, rstd = ctx.saved_tensors
        word, pos, typ, gamma, beta, type_index, pad_idx = ctx.params
        B, L = ids.shape
        H = z.shape[-1]
        rows = B * L
        dy = dy.contiguous()
        dz = torch.empty_like(dy)
        ws = RT.workspace(dy.device, lib.load().bevbert_colsum_workspace_floats(3 * H))
        sg, sb = _sink(gamma), _sink(beta)
        assert (sg is None) == (sb is None)
        # the broadcast token-type row: its gradient is the column sum of dz = the kernel's third (dbias) output, through
        # the deterministic two-stage reduction (a torch sum over the 5 120 rows would depend on the GPU's load)
        styp = _sink(typ)[type_index] if (sg is not None and typ.requires_grad and _sink(typ) is not None) else None
        if sg is not None:
            _mark_touched(gamma); _mark_touched(beta)
            if styp is not None:
                _mark_touched(typ)
            if WgradStream.DEFER_FINALIZE:
                nb = _partial_rows(rows)
                part = RT.scratch.alloc(nb * 3 * H * 4, dy.device)
                call("bevbert_layernorm_bwd", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(_f32(gamma)), ptr(dz), None,
                     None, None, None, part, rows, H, dtype_code(dy), 0.0, 0, 0, 1, stream())
                ReduceQueue.add(part, nb, 3, H, (ptr(sg), ptr(sb), ptr(styp)))
            else:
                call("bevbert_layernorm_bwd", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(_f32(gamma)), ptr(dz), None,
                     ptr(sg), ptr(sb), ptr(styp), ptr(ws), rows, H, dtype_code(dy), 0.0, 0, 0, 1, stream())
            rg = rb = None
        else:
            rg = torch.empty(H, dtype=torch.float32, device=dy.device)
            rb = torch.empty(H, dtype=torch.float32, device=dy.device)
            call("bevbert_layernorm_bwd", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(_f32(gamma)), ptr(dz), None,
                 ptr(rg), ptr(rb), None, ptr(ws), rows, H, dtype_code(dy), 0.0, 0, 0, 0, stream())
        dz2 = dz.reshape(rows, H)
        dzf = dz2          # (kept alive by the deferred closures below)

        def word_grad(t):
            call("bevbert_embedding_grad", ptr(ids), ptr(dz2), ptr(t), rows, H, pad_idx, dtype_code(dz2), stream())

        def pos_grad(t):        # position p receives the sum over the batch: column sums of dz viewed as (B, L * H)
            call("bevbert_colsum_any", ptr(dz2), ptr(t), B, L * H, dtype_code(dz2), 1, stream())

        def typ_grad(t):        # (only when the LayerNorm backward's third column sum could not take it, see styp)
            call("bevbert_colsum_any", ptr(dz2), ptr(t[type_index]), rows, H, dtype_code(dz2), 1, stream())

        makers = ((word, word_grad), (pos, pos_grad), (typ, typ_grad))
        outs, deferred = [], []
        for p, make in makers:
            if not p.requires_grad:
                outs.append(None)
            elif p is typ and styp is not None:
                outs.append(None)              # written by the LayerNorm backward's column reduction above
            elif _sink(p) is not None:
                _mark_touched(p)
                deferred.append((make, _sink(p)))
                outs.append(None)
            else:
                t = torch.zeros(p.shape, dtype=torch.float32, device=dy.device)
                make(t)
                outs.append(t.to(p.dtype))
        if deferred:
            # every write into a parameter's gradient sink goes through the weight-gradient stream: the word table also
            # receives the tied MLM decoder's dW there (a deferred, non-atomic read-modify-write), the type table the
            # panorama branch's row-1 gradient -- one stream keeps the writers of a sink in program order
            WgradStream.submit(dy.device, lambda: [m(t) for m, t in deferred], dz2, dzf, ids, dy)
        return (None, outs[0], outs[1], outs[2], rg, rb, None, None, None, None, None, None)


def embed_sum_layernorm(ids, word, pos, typ, gamma, beta, eps, type_index=0, padding_idx=None):
    """BertEmbeddings (vilmodel.py:62-77): LN(word[ids] + pos[0..L) + type[type_index]).  ``padding_idx``: rows of the
    word table that receive no lookup gradient (nn.Embedding(padding_idx=0), vilmodel.py:50)."""
    return _EmbedLN.apply(ids, word, pos, typ, gamma, beta, eps, type_index, _compute(word), _compute(pos),
                          _compute(typ), -1 if padding_idx is None else int(padding_idx))


# ----------------------------------------------------------------------------- K6 segment gather
class SegmentCSR:
    """Host-built CSR (and its transpose) describing out[r] = sum_e w[e] * src[idx[e]].

    ``capacity`` (entries) fixes the size of the device arrays, so that a later batch of the same shape bucket can be
    written into the SAME buffers (``update``) -- the kernels only read the ranges the row pointers describe."""

    def __init__(self, rowptr, idx, w, n_src, device, capacity=None):
        self.n_out, self.n_src = len(rowptr) - 1, int(n_src)
        self.capacity = int(capacity) if capacity is not None else len(idx)
        pack, packw = self._pack(rowptr, idx, w)
        di = torch.from_numpy(pack).to(device, non_blocking=True)
        dw = torch.from_numpy(packw).to(device, non_blocking=True)
        self._di, self._dw = di, dw
        n0, n1, n2 = self.n_out + 1, self.capacity, self.n_src + 1
        self.rowptr, self.idx = di[:n0], di[n0:n0 + n1]
        self.t_rowptr, self.t_idx = di[n0 + n1:n0 + n1 + n2], di[n0 + n1 + n2:]
        self.w, self.t_w = dw[:n1], dw[n1:]

    def _pack(self, rowptr, idx, w):
        import numpy as np
        rowptr = np.asarray(rowptr, dtype=np.int32)
        idx = np.asarray(idx, dtype=np.int32)
        w = np.asarray(w, dtype=np.float32)
        assert len(rowptr) == self.n_out + 1 and len(idx) <= self.capacity, "segment CSR does not fit its buffers"
        # transpose: for each src row, the (out row, weight) pairs that read it
        out_of_e = np.repeat(np.arange(self.n_out, dtype=np.int32), np.diff(rowptr))
        order = np.argsort(idx, kind="stable")
        t_rowptr = np.zeros(self.n_src + 1, dtype=np.int32)
        np.add.at(t_rowptr, idx + 1, 1)
        t_rowptr = np.cumsum(t_rowptr).astype(np.int32)
        pad = np.zeros(self.capacity - len(idx), dtype=np.int32)
        padw = pad.astype(np.float32)
        pack = np.concatenate([rowptr, idx, pad, t_rowptr, out_of_e[order], pad]).astype(np.int32)
        packw = np.concatenate([w, padw, w[order], padw]).astype(np.float32)
        return pack, packw

    def update(self, rowptr, idx, w):
        """Write another aggregation of the same shape (rows, sources, <= capacity entries) into the device arrays."""
        pack, packw = self._pack(rowptr, idx, w)
        self._di.copy_(torch.from_numpy(pack), non_blocking=True)
        self._dw.copy_(torch.from_numpy(packw), non_blocking=True)


class _SegmentWsum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, csr):
        assert src.is_contiguous() and src.dim() == 2 and src.shape[0] == csr.n_src
        out = torch.empty(csr.n_out, src.shape[1], dtype=src.dtype, device=src.device)
        call("bevbert_segment_wsum", ptr(src), ptr(csr.rowptr), ptr(csr.idx), ptr(csr.w), ptr(out), csr.n_out,
             src.shape[1], dtype_code(src), stream())
        ctx.csr = csr
        return out

    @staticmethod
    def backward(ctx, dout):
        csr = ctx.csr
        dout = dout.contiguous()
        dsrc = torch.empty(csr.n_src, dout.shape[1], dtype=dout.dtype, device=dout.device)
        call("bevbert_segment_wsum", ptr(dout), ptr(csr.t_rowptr), ptr(csr.t_idx), ptr(csr.t_w), ptr(dsrc), csr.n_src,
             dout.shape[1], dtype_code(dout), stream())
        return dsrc, None


def segment_wsum(src, csr):
    return _SegmentWsum.apply(src, csr)


# ----------------------------------------------------------------------------- K1 BEV splat (no gradient)
def pixel_scale(hw, device, vfov=math.radians(90)):
    """((u + .5 - c) / f) in fp32 exactly as bev_utils.py:91-137 builds it (f = hw / (2 tan(vfov/2)), c = hw/2)."""
    f = torch.tensor(hw / (2.0 * math.tan(vfov / 2.0)), dtype=torch.float32)
    c = torch.tensor(hw / 2.0, dtype=torch.float32)
    return ((torch.arange(hw, dtype=torch.float32) + 0.5 - c) / f).to(device)


@torch.no_grad()
def bev_lift_bin(depths, T_c2w, T_w2c, S_w2c, pix, dim, res, depth_scale=10.0, y_clip=0.5):
    B, V = depths.shape[0], depths.shape[1]
    hw = depths.shape[-1]
    P = V * hw * hw
    dev = depths.device
    cell = torch.empty(B, P, dtype=torch.int32, device=dev)
    order = torch.empty(B, P, dtype=torch.int32, device=dev)
    call("bevbert_zero", ptr(order), order.numel() * 4, stream())       # (dropped points leave their slots unwritten)
    cell_start = torch.empty(B, dim * dim + 1, dtype=torch.int32, device=dev)
    f = lambda t: t.contiguous().float()
    d, a, b_, c_ = f(depths), f(T_c2w), f(T_w2c), f(S_w2c)
    call("bevbert_bev_lift_bin", ptr(d), ptr(a), ptr(b_), ptr(c_), ptr(pix), B, V, hw, float(depth_scale), dim,
         float(res), float(y_clip), ptr(cell), ptr(order), ptr(cell_start), stream())
    return cell, order, cell_start


@torch.no_grad()
def bev_bin_points(points, drop_mask, dim, res, y_clip=0.5):
    B, P = points.shape[0], points.shape[1]
    dev = points.device
    cell = torch.empty(B, P, dtype=torch.int32, device=dev)
    order = torch.zeros(B, P, dtype=torch.int32, device=dev)
    cell_start = torch.empty(B, dim * dim + 1, dtype=torch.int32, device=dev)
    pts = points.contiguous().float()
    dm = drop_mask.contiguous().to(torch.uint8)
    call("bevbert_bev_bin_points", ptr(pts), ptr(dm), B, P, dim, float(res), float(y_clip), ptr(cell), ptr(order),
         ptr(cell_start), stream())
    return cell, order, cell_start


@torch.no_grad()
def bev_splat_mean(feat, order, cell_start, K, out_dtype=None, sems=None, n_classes=40, rows=None):
    """feat (B,P,C) f32/bf16/f16 -> (B,K,C); sems: (B,P) uint8 ids or (B,P,S) float64 one-hot or None.
    rows (B,) or (B,R) int32: feat / sems are (N,P0,...) stores (feature_store.GridFeatureStore) and sample b's points
    are the R store rows rows[b] back to back (P = R * P0 = order.shape[1])."""
    C = feat.shape[-1]
    B, P = order.shape
    R = 1 if rows is None or rows.dim() == 1 else rows.shape[1]
    assert P == R * feat.shape[1] if rows is not None else feat.shape[:2] == (B, P)
    assert feat.is_contiguous() if rows is not None else True
    feat = feat.contiguous()
    out_dtype = out_dtype or (feat.dtype if feat.dtype != torch.float16 else torch.float32)
    out = torch.empty(B, K, C, dtype=out_dtype, device=feat.device)
    sem_ids = sem_dense = out_sem = out_mask = None
    S = n_classes
    if sems is not None:
        if sems.dim() == 2:
            sem_ids = sems.contiguous().to(torch.uint8)
            assert sem_ids.shape[1] * R == P and (rows is not None or sem_ids.shape[0] == B)
        else:
            sem_dense = sems.contiguous().to(torch.float64)
            S = sems.shape[-1]
        out_sem = torch.empty(B, K, S, dtype=torch.uint8, device=feat.device)
        out_mask = torch.empty(B, K, dtype=torch.bool, device=feat.device)     # the kernel writes the bytes 0 / 1
    call("bevbert_bev_splat_mean", ptr(feat), dtype_code(feat), ptr(order), ptr(cell_start), ptr(out),
         dtype_code(out_dtype), B, P, K, C, ptr(sem_ids), ptr(sem_dense), S, ptr(out_sem), ptr(out_mask),
         ptr(rows.contiguous() if rows is not None else None), R, stream())
    return out, out_sem, out_mask


def dropout_keep_mask(n, drop_p, seed, offset, device):
    out = torch.empty(n, dtype=torch.uint8, device=device)
    call("bevbert_dropout_keep_mask", ptr(out), n, float(drop_p), int(seed), int(offset), stream())
    return out.bool()
