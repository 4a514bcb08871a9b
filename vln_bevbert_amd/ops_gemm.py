"""Library GEMMs of the Linear layers (ops.py re-exports everything here): hipBLASLt through the C ABI with cached
per-problem plans and the shipped choice table, split-K weight gradients into the arena, and the autograd Functions of
``nn.Linear`` in its plain, packed (Q|K|V) and hoisted (all K|V projections of an encoder) forms."""
import os as _os

import torch
import torch.nn.functional as F

from . import lib
from .lib import dtype_code, ptr, stream
from .ops_core import RT, _LT_WS_BYTES, _compute, _gemm, _mark_touched, _sink, call
from .ops_reduce import ReduceQueue, WgradStream, _on_launch_stream, _partial_rows


# Library GEMMs go straight to hipBLASLt through the C ABI (bevbert_gemm): ~7 us of host time per call instead of the
# ~28 us of torch.mm / F.linear dispatch -- the training step is host-bound at batch 64 (bench.py reports both clocks).
# BEVBERT_LT_GEMM=0 routes them through torch instead (same library underneath); A/B knob.
_LT_ENABLED = _os.environ.get("BEVBERT_LT_GEMM", "1") == "1"


_LT_AUTOTUNE = int(_os.environ.get("BEVBERT_LT_AUTOTUNE", "32"))


_LT_UNSUPPORTED = set()


# Every new problem costs one timing pass (32 candidates x 10 launches + a sync) the first time it is seen.  The R2R step
# has ~100 problems; real batches add data-dependent row counts (masked tokens, selected cells, trajectory lengths).
# Past this many plans new problems stay on torch's own GEMM path (the library's single heuristic pick, no timing pass)
# so that an unbounded variety of shapes cannot turn into an unbounded number of stalls.
_LT_PLAN_BUDGET = int(_os.environ.get("BEVBERT_LT_PLAN_BUDGET", "8192"))


_LT_PLANS = {}


GEMM_FALLBACKS = {}      # (kind, M, N, K) -> calls that went through torch's GEMM instead of the direct hipBLASLt path


def _warn_fallback(kind, M, N, K, why="no direct hipBLASLt plan"):
    key = (kind, int(M), int(N), int(K))
    n = GEMM_FALLBACKS.get(key, 0)
    GEMM_FALLBACKS[key] = n + 1
    if n == 0:
        import warnings
        warnings.warn(f"vln_bevbert_amd: {kind} GEMM M={M} N={N} K={K} runs through torch ({why}); same library, "
                      f"~4x the host cost per call", RuntimeWarning, stacklevel=3)


GEMM_TUNING_FILE = _os.environ.get("BEVBERT_GEMM_TABLE",
                                   _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "gemm_tuning.txt"))


_tuning_loaded = False


def load_gemm_tuning_table(path=None):
    """Import the shipped hipBLASLt choice table (bevbert_gemm_tuning_import); returns the number of rows (0 when the
    file is missing or was made with another library version -- the plans then time their candidates on first use)."""
    global _tuning_loaded
    _tuning_loaded = True
    path = path or GEMM_TUNING_FILE
    if not _os.path.exists(path):
        return 0
    with open(path, "rb") as f:
        n = lib.load().bevbert_gemm_tuning_import(f.read())
    return max(n, 0)


def save_gemm_tuning_table(path):
    """Write the choices of every plan tuned so far (plus the imported rows) for later runs."""
    l = lib.load()
    need = l.bevbert_gemm_tuning_export(None, 0)
    import ctypes
    buf = ctypes.create_string_buffer(need)
    l.bevbert_gemm_tuning_export(buf, need)
    with open(path, "wb") as f:
        f.write(buf.value)
    return buf.value.count(b"\n") - 1


def _lt_gemm(a, b, out, bias, M, N, K, opA, opB, lda, ldb, ldc, batch=1, sa=0, sb=0, sc=0, accumulate=0, c_in=None):
    """out (+)= op(a) . op(b) (+ bias) on hipBLASLt via the C ABI; False if the library has no kernel for the shape.
    ``c_in`` (with accumulate=1): out = product + c_in, the addend being a separate buffer of out's layout."""
    key = (M, N, K, opA, opB, lda, ldb, ldc, batch, a.dtype, out.dtype, None if bias is None else bias.dtype,
           accumulate)
    plan = _LT_PLANS.get(key)
    if plan is None:
        if len(_LT_PLANS) >= _LT_PLAN_BUDGET:
            return False
        if not _tuning_loaded:
            load_gemm_tuning_table()
        plan = lib.load().bevbert_gemm_plan(M, N, K, opA, opB, lda, ldb, ldc, batch, sa, sb, sc, dtype_code(a),
                                            dtype_code(out), -1 if bias is None else dtype_code(bias), accumulate,
                                            _LT_WS_BYTES, _LT_AUTOTUNE)
        if plan < 0:
            raise lib.BevBertHipError(f"bevbert_gemm_plan failed ({plan}): {lib.load().bevbert_last_error().decode()}")
        _LT_PLANS[key] = plan
    if plan in _LT_UNSUPPORTED:
        return False
    st = stream()
    if c_in is not None:
        rc = lib.load().bevbert_gemm_run_add(plan, a.data_ptr(), b.data_ptr(), c_in.data_ptr(), out.data_ptr(),
                                             None if bias is None else bias.data_ptr(),
                                             RT.gemm_workspace(a.device, st), _LT_WS_BYTES, st)
    else:
        rc = _LT_RUN(plan, a.data_ptr(), b.data_ptr(), out.data_ptr(), None if bias is None else bias.data_ptr(),
                     RT.gemm_workspace(a.device, st), _LT_WS_BYTES, st)
    if rc == -3:
        _LT_UNSUPPORTED.add(plan)
        return False
    if rc != 0:
        raise lib.BevBertHipError(f"bevbert_gemm_run failed ({rc}): {lib.load().bevbert_last_error().decode()}")
    return True


def _LT_RUN(*args):
    global _LT_RUN
    _LT_RUN = lib.load().bevbert_gemm_run        # bind once; later calls go straight to the ctypes function
    return _LT_RUN(*args)


def _rows(t):
    """2-D row-major view (rows, C) of a tensor with unit inner stride and its row stride."""
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(1) != 1 or (t2.shape[0] > 1 and t2.stride(0) < t2.shape[1]):
        t2 = t2.contiguous()
    return t2, (t2.stride(0) if t2.shape[0] > 1 else t2.shape[1])


def _lt_ok(*ts):
    return _LT_ENABLED and all(t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) for t in ts)


def _linear_fwd(x, w_c, b_c):
    """y = x w_c^T (+ b_c)."""
    N, K = w_c.shape
    if _lt_ok(x, w_c) and x.dtype == w_c.dtype and w_c.stride(1) == 1 and x.numel() > 0:
        x2, lda = _rows(x)
        M = x2.shape[0]
        y = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
        if _lt_gemm(x2, w_c, y, b_c, M, N, K, 0, 1, lda, w_c.stride(0), N):
            return y
    if x.is_cuda:
        _warn_fallback("fwd", x.numel() // max(1, K), N, K)
    return F.linear(x, w_c, b_c)


def _linear_dgrad(dy2, w_c, add=None):
    """dx (M x K) = dy2 (M x N) w_c (N x K) (+ add, an (M x K) tensor folded in as the GEMM's beta = 1 addend)."""
    N, K = w_c.shape
    if add is not None:
        add = add.reshape(-1, K)
        if not add.is_contiguous() or add.dtype != dy2.dtype:
            add = add.to(dy2.dtype).contiguous()
    if _lt_ok(dy2, w_c) and dy2.dtype == w_c.dtype and w_c.stride(1) == 1 and dy2.numel() > 0:
        d2, lda = _rows(dy2)
        M = d2.shape[0]
        dx = torch.empty(M, K, dtype=dy2.dtype, device=dy2.device)
        if add is None:
            if _lt_gemm(d2, w_c, dx, None, M, K, N, 0, 0, lda, w_c.stride(0), K):
                return dx
        elif _lt_gemm(d2, w_c, dx, None, M, K, N, 0, 0, lda, w_c.stride(0), K, accumulate=1, c_in=add):
            return dx
    if dy2.is_cuda:
        _warn_fallback("dgrad", dy2.shape[0], K, N)
    return dy2.mm(w_c) if add is None else torch.addmm(add, dy2, w_c)


def _linear_wgrad(dy2, x2, S=1, scratch=False):
    """(S x) N x K partial products dy2^T x2 over S equal chunks of the token axis (compute dtype); ``scratch``: the
    product lives in the scratch ring (it is consumed by the batched accumulate at the end of the backward pass)."""
    M, N = dy2.shape
    K = x2.shape[1]
    if _lt_ok(dy2, x2) and dy2.dtype == x2.dtype and M > 0:
        d2, lda = _rows(dy2)
        xx, ldb = _rows(x2)
        if S == 1 or (lda == N and ldb == K):
            shape = (S, N, K) if S > 1 else (N, K)
            part = RT.scratch.tensor(shape, dy2.dtype, dy2.device) if scratch else \
                torch.empty(shape, dtype=dy2.dtype, device=dy2.device)
            Ms = M // S
            if _lt_gemm(d2, xx, part, None, N, K, Ms, 1, 0, lda, ldb, K, S, Ms * lda, Ms * ldb, N * K):
                return part
    if dy2.is_cuda:
        _warn_fallback("wgrad", N, K, M)
    if S > 1:
        return _on_launch_stream(lambda: torch.bmm(dy2.view(S, M // S, N).transpose(1, 2), x2.view(S, M // S, K)))
    return _on_launch_stream(lambda: dy2.t().mm(x2))


_SPLITK_ENABLED = _os.environ.get("BEVBERT_SPLITK", "1") == "1"     # A/B knob


_SPLITK_MAX = int(_os.environ.get("BEVBERT_SPLITK_MAX", "16"))


def _split_k(M, N, K):
    """Number of token-axis chunks for a weight-gradient GEMM dW(N x K) = dy^T(N x M) x(M x K).

    The output is small (9..36 tiles of 256x256) and the reduction axis M is long (5 120 .. 28 224 tokens), so a plain
    GEMM leaves most of the 256 CUs idle (measured 140-250 TFLOP/s); a batched GEMM over S chunks of M fills them
    (600-880 TFLOP/s, scripts/bench_wgrad.py).  Aim at 144-256 workgroups, keep >= 640 tokens per chunk."""
    if not _SPLITK_ENABLED:
        return 1
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    s = 1
    while s * 2 <= min(_SPLITK_MAX, M // 640) and s * 2 * tiles <= 256 and M % (s * 2) == 0:
        s *= 2
    return s


def _wgrad_into(sink, dy2, x2):
    """sink (fp32 arena view, N x K) += dy2^T @ x2 with host-side split-K and a fused partial-sum + accumulate."""
    M, N = dy2.shape
    K = x2.shape[1]
    if dy2.dtype == torch.float32 and not (_LT_ENABLED and dy2.is_cuda):
        if dy2.is_cuda:
            _warn_fallback("wgrad", N, K, M, "BEVBERT_LT_GEMM=0")
        _gemm("wgrad", lambda: _on_launch_stream(lambda: sink.addmm_(dy2.t(), x2)), N, K, M)
        return None
    S = _split_k(M, N, K) if dy2.dtype != torch.float32 else 1
    if not (S > 1 and dy2.is_contiguous() and x2.is_contiguous()):
        S = 1
    batched = WgradStream.DEFER_FINALIZE and (N * K) % 4 == 0 and dy2.is_cuda
    part = _gemm("wgrad", lambda: _linear_wgrad(dy2, x2, S, scratch=batched), N, K, M)
    if batched:           # folded into the arena by ONE launch per backward pass, together with every other weight's
        ReduceQueue.add_accum(part.data_ptr(), sink.data_ptr(), S, N * K, dtype_code(part))
        WgradStream._keep.append(part)     # (a product that came from torch's fallback GEMM must outlive the flush)
        return None
    if (N * K) % 4 == 0:
        call("bevbert_accum_partials", ptr(part), ptr(sink), S, N * K, dtype_code(part), stream())
    else:
        _on_launch_stream(lambda: sink.add_(part if S == 1 else part.sum(0)))
    return part


def _param_grads(w_sink, b_sink, dyc, xc):
    """dW += dy^T x and db += colsum(dy) into the gradient arena (the deferred body of a Linear's backward)."""
    if w_sink is not None:
        part = _wgrad_into(w_sink, dyc, xc)
        if part is not None and lib._override is not None:
            WgradStream._keep.append(part)
    if b_sink is not None:
        C = dyc.shape[1]
        if WgradStream.DEFER_FINALIZE and dyc.shape[0] > 0:
            nb = _partial_rows(dyc.shape[0])
            part = RT.scratch.alloc(nb * C * 4, dyc.device)
            call("bevbert_colsum_partials", ptr(dyc), part, dyc.shape[0], C, dtype_code(dyc), stream())
            ReduceQueue.add(part, nb, 1, C, (ptr(b_sink), None, None))
        else:
            ws = RT.workspace(dyc.device, lib.load().bevbert_colsum_workspace_floats(C))
            call("bevbert_colsum", ptr(dyc), ptr(b_sink), ptr(ws), dyc.shape[0], C, dtype_code(dyc), 1, stream())


# ----------------------------------------------------------------------------- library GEMM with arena wgrad
class _Linear(torch.autograd.Function):
    """y = x W^T (+ b) on hipBLASLt; backward writes dW / db straight into the gradient arena."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_c, b_c, tap=False):
        y = _gemm("fwd", lambda: _linear_fwd(x, w_c, b_c), x.numel() // x.shape[-1], w_c.shape[0], w_c.shape[1])
        ctx.save_for_backward(x, w_c)
        ctx.params = (weight, bias)
        ctx.tap = tap
        # tap: the input ALSO feeds a residual connection.  It is handed back as a second output, so that the residual's
        # gradient arrives HERE and is folded into the input-gradient GEMM (dx = dy W + d_res, beta = 1) -- autograd would
        # otherwise add the two gradients of x with a separate elementwise kernel (~50 of them per training step)
        ctx.set_materialize_grads(False)      # (y, tap): an unused output arrives as None, not as dense zeros
        return (y, x.view_as(x)) if tap else y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, w_c = ctx.saved_tensors
        weight, bias = ctx.params
        if dy is None:                      # only the residual tap carried a gradient
            return dres, None, None, None, None, None
        dy2 = dy.reshape(-1, dy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        M, N, K = dy2.shape[0], dy2.shape[1], x2.shape[1]
        dx = _gemm("dgrad", lambda: _linear_dgrad(dy2, w_c, dres), M, K, N).view(x.shape) if ctx.needs_input_grad[0] else None
        gw = gb = None
        w_sink = _sink(weight) if weight.requires_grad else None
        b_sink = _sink(bias) if (bias is not None and bias.requires_grad) else None
        C = dy2.shape[1]
        if weight.requires_grad and w_sink is None:
            gw = _linear_wgrad(dy2, x2).to(weight.dtype)
        if bias is not None and bias.requires_grad and (b_sink is None or C % 4 != 0):
            if C % 4 != 0:                                  # the 1-wide heads, the 30 522-wide vocabulary bias
                dyc = dy2 if dy2.is_contiguous() else dy2.contiguous()
                if b_sink is not None and dyc.is_cuda:
                    _mark_touched(bias)
                    call("bevbert_colsum_any", ptr(dyc), ptr(b_sink), dyc.shape[0], C, dtype_code(dyc), 1, stream())
                    b_sink = None
                elif b_sink is not None:
                    _mark_touched(bias)
                    b_sink.add_(dy2.float().sum(0))
                    b_sink = None
                else:
                    gb = dy2.float().sum(0).to(bias.dtype)
            else:
                ws = RT.workspace(dy.device, lib.load().bevbert_colsum_workspace_floats(C))
                dyc = dy2 if dy2.is_contiguous() else dy2.contiguous()
                t = torch.empty(C, dtype=torch.float32, device=dy.device)
                call("bevbert_colsum", ptr(dyc), ptr(t), ptr(ws), dyc.shape[0], C, dtype_code(dyc), 0, stream())
                gb = t.to(bias.dtype)
        if w_sink is not None or b_sink is not None:        # arena parameters: accumulate on the weight-gradient stream
            dyc = dy2 if dy2.is_contiguous() else dy2.contiguous()
            xc = x2 if x2.is_contiguous() else x2.contiguous()
            if w_sink is not None:
                _mark_touched(weight)
            if b_sink is not None:
                _mark_touched(bias)
            WgradStream.submit(dy.device, lambda: _param_grads(w_sink, b_sink, dyc, xc), dyc, xc, dy)
        return dx, gw, gb, None, None, None


def linear(x, weight, bias=None, w_c=None, b_c=None):
    """F.linear with compute-dtype weights; ``weight``/``bias`` are the master parameters (gradient owners)."""
    if w_c is None:
        w_c = _compute(weight)
    if bias is not None and b_c is None:
        b_c = _compute(bias)
    return _Linear.apply(x, weight, bias, w_c, b_c)


def linear_res(x, weight, bias=None):
    """(linear(x), x) for an input that also feeds a residual connection: use the SECOND output as the residual and
    the gradient of the residual branch is folded into this layer's input-gradient GEMM (see _Linear.forward)."""
    if not (x.requires_grad and torch.is_grad_enabled()) or getattr(x, "_res32", None) is not None:
        # (fp32 residual stream: the residual is x's fp32 twin, its gradient joins x's inside the LayerNorm backward kernel)
        return linear(x, weight, bias), x
    return _Linear.apply(x, weight, bias, _compute(weight), None if bias is None else _compute(bias), True)


class _PackedParam:
    """A contiguous run of arena parameters used as one GEMM operand (packed QKV / KV projections)."""

    def __init__(self, params, compute, main_grad):
        self.params, self.compute, self.main_grad = params, compute, main_grad
        self.requires_grad = any(p.requires_grad for p in params)
        self.dtype = params[0].dtype
        self.arena = getattr(params[0], "arena", None)

    def touch(self):
        for p in self.params:
            _mark_touched(p)


class _LinearPacked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pw, pb, tap=False):
        ctx.save_for_backward(x)
        ctx.packed = (pw, pb)
        y = _gemm("fwd", lambda: _linear_fwd(x, pw.compute, pb.compute), x.numel() // x.shape[-1],
                  pw.compute.shape[0], pw.compute.shape[1])
        ctx.set_materialize_grads(False)
        return (y, x.view_as(x)) if tap else y          # residual tap: see _Linear.forward

    @staticmethod
    def backward(ctx, dy, dres=None):
        (x,) = ctx.saved_tensors
        pw, pb = ctx.packed
        if dy is None:
            return dres, None, None, None
        dy2 = dy.reshape(-1, dy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        M, N, K = dy2.shape[0], dy2.shape[1], x2.shape[1]
        dx = _gemm("dgrad", lambda: _linear_dgrad(dy2, pw.compute, dres), M, K, N).view(x.shape) if ctx.needs_input_grad[0] else None
        if pw.requires_grad:
            pw.touch()
            pb.touch()
            C = dy2.shape[1]
            dyc = dy2 if dy2.is_contiguous() else dy2.contiguous()
            xc = x2 if x2.is_contiguous() else x2.contiguous()
            WgradStream.submit(dy.device, lambda: _param_grads(pw.main_grad, pb.main_grad, dyc, xc), dyc, xc, dy)
        return dx, None, None, None


def linear_packed(x, pw, pb):
    return _LinearPacked.apply(x, pw, pb)


HOIST_KV = _os.environ.get("BEVBERT_HOIST_KV", "1") == "1"      # A/B knob


class _KVGradHolder:
    """The (B, Lk, layers * 2H) gradient buffer of a hoisted K/V projection, allocated when the first attention backward
    asks for its slice."""

    def __init__(self, n_layers, width):
        self.n, self.width, self.buf = n_layers, width, None

    def grad_slice(self, layer, like):
        if self.buf is None:
            self.buf = torch.empty(like.shape[:-1] + (self.n * self.width,), dtype=like.dtype, device=like.device)
        return self.buf[..., layer * self.width:(layer + 1) * self.width]


class _HoistedKV(torch.autograd.Function):
    """The key / value projections of ALL cross-attention layers of an encoder in one GEMM.

    The context of the cross-attention is the same tensor in every layer (the text states in the map encoders --
    ``lang_feats`` is never updated, vilmodel.py:383-398,446-463 -- or the BEV / map tokens in the MLM direction), so
    layers x (x W_kv^T) is one (rows, layers * 2H, C) problem: at the 5 120 text rows of the step that is 96 output
    tiles of 256 x 256 instead of four launches of 24.  Backward: each layer's attention writes dK / dV into its column
    slice of one buffer (``_KVGradHolder``); when the last one has run, ONE K-concatenated input-gradient GEMM
    (rows x C, K = layers * 2H) and ONE weight-gradient GEMM (layers * 2H x C) into the arena follow."""

    @staticmethod
    def forward(ctx, x, pw, pb, n_layers):
        ctx.save_for_backward(x)
        ctx.packed = (pw, pb)
        y = _gemm("fwd", lambda: _linear_fwd(x, pw.compute, pb.compute), x.numel() // x.shape[-1],
                  pw.compute.shape[0], pw.compute.shape[1])
        width = y.shape[-1] // n_layers
        ctx.holder = _KVGradHolder(n_layers, width)
        return tuple(y[..., i * width:(i + 1) * width] for i in range(n_layers))

    @staticmethod
    def backward(ctx, *grads):
        (x,) = ctx.saved_tensors
        pw, pb = ctx.packed
        h = ctx.holder
        some = next((g for g in grads if g is not None), None)
        if some is None:
            return None, None, None, None
        for i, g in enumerate(grads):
            dst = h.grad_slice(i, some)
            if g is None:
                dst.zero_()                        # a layer whose output reached no loss
            elif g.data_ptr() != dst.data_ptr() or g.stride() != dst.stride():
                dst.copy_(g)                       # a gradient that did not come from the attention backward (tests)
        dy2 = h.buf.reshape(-1, h.buf.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        M, N, K = dy2.shape[0], dy2.shape[1], x2.shape[1]
        dx = _gemm("dgrad", lambda: _linear_dgrad(dy2, pw.compute), M, K, N).view(x.shape) if ctx.needs_input_grad[0] else None
        if pw.requires_grad:
            pw.touch()
            pb.touch()
            xc = x2 if x2.is_contiguous() else x2.contiguous()
            WgradStream.submit(dy2.device, lambda: _param_grads(pw.main_grad, pb.main_grad, dy2, xc), dy2, xc, h.buf)
        return dx, None, None, None


def hoisted_kv(context, pw, pb, n_layers):
    """[(B, Lk, 2H) K|V view of layer i] for the cross-attention layers whose packed parameters ``pw`` (layers * 2H, C) /
    ``pb`` (layers * 2H) describe; pass the views as ``kv=`` to BertOutAttention.forward."""
    outs = _HoistedKV.apply(context, pw, pb, n_layers)
    if torch.is_grad_enabled() and any(o.requires_grad for o in outs):
        holder = outs[0].grad_fn.holder if hasattr(outs[0].grad_fn, "holder") else None
        if holder is not None:
            for i, o in enumerate(outs):
                o._kv_grad_slot = (holder, i)
    return outs


def linear_packed_res(x, pw, pb):
    """(packed projection of x, x as residual tap) -- see linear_res."""
    if not (x.requires_grad and torch.is_grad_enabled()) or getattr(x, "_res32", None) is not None:
        return _LinearPacked.apply(x, pw, pb), x
    return _LinearPacked.apply(x, pw, pb, True)


def gemm_plan_count():
    """Number of hipBLASLt plans the library holds (grows when a new GEMM problem shows up: a new shape bucket)."""
    return int(lib.load().bevbert_gemm_plan_count())
