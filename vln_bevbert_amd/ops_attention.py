"""Attention entry points (ops.py re-exports everything here): one autograd Function over bevbert_attn_fwd / _bwd for
self-, cross- and graph-biased attention, with the keep-bit workspace of the dropout mask."""
import math
import os as _os

import torch

from .lib import dtype_code, ptr, stream
from .ops_core import ATTN_BITS, HEAD_DIM, RT, _drop_bits_words, _mark_touched, _sink, call


# ----------------------------------------------------------------------------- K2 attention
def _strides(q, k, v, o):
    for t in (q, k, v, o):
        assert t.dim() == 3 and t.stride(2) == 1, "attention operands are (B, L, nh*64) with unit inner stride"
    import ctypes
    arr = (ctypes.c_int64 * 8)(q.stride(1), k.stride(1), v.stride(1), o.stride(1),
                               q.stride(0), k.stride(0), v.stride(0), o.stride(0))
    return arr


class _Attention(torch.autograd.Function):
    """mode 'self': qkv packed (B,L,3H);  mode 'cross': q (B,Lq,H) + kv packed (B,Lk,2H);  mode 'sep': q,k,v."""

    @staticmethod
    def forward(ctx, mode, a, b_, c_, key_mask, bias, nh, drop_p, impl):
        if mode == "self":
            H = a.shape[-1] // 3
            q, k, v = a[..., :H], a[..., H:2 * H], a[..., 2 * H:]
        elif mode == "cross":
            H = a.shape[-1]
            q, k, v = a, b_[..., :H], b_[..., H:]
        else:
            H = a.shape[-1]
            q, k, v = a, b_, c_
        assert H == nh * HEAD_DIM
        B, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]
        o = torch.empty(B, Lq, H, dtype=q.dtype, device=q.device)
        need_grad = any(ctx.needs_input_grad)
        lse = torch.empty(B, nh, Lq, dtype=torch.float32, device=q.device) if need_grad else None
        off = RT.next_offset(B * nh * Lq * Lk) if drop_p > 0 else 0
        scale = 1.0 / math.sqrt(HEAD_DIM)
        if key_mask is not None:
            assert key_mask.dtype == torch.float32 and key_mask.shape == (B, Lk) and key_mask.is_contiguous()
        if bias is not None:
            assert bias.dtype == torch.float32 and bias.shape == (B, Lq, Lk) and bias.is_contiguous()
        bits, bits_ready = None, 0
        if drop_p > 0 and q.dtype == torch.bfloat16 and impl != 1:
            # keep-bit workspace of the dropout mask (1 bit / element in the forward's and in the backward's lane
            # layout: 2 x 19 MB at 64x12x441x441), filled by the library ahead of the forward kernel; both directions
            # read bits through the scalar cache instead of hashing per element
            # attn_small.hip (opt-in, BEVBERT_ATTN_SMALL=1) hashes inline whatever the query count
            short_keys = Lk <= 96 and bias is None and _os.environ.get("BEVBERT_ATTN_SMALL") == "1"
            if (Lq * Lk >= 32768 or Lk > 256) and not short_keys:
                bits, bits_ready = ATTN_BITS.get(B, nh, Lq, Lk, drop_p, off, q.device)
            else:       # small score matrices: the forward hashes inline and leaves the bits for the backward (capi.hip)
                bits = torch.empty(_drop_bits_words(B, nh, Lq, Lk), dtype=torch.int64, device=q.device)
        call("bevbert_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(key_mask), ptr(bias),
             _strides(q, k, v, o), B, nh, Lq, Lk, HEAD_DIM, scale, dtype_code(q), impl, float(drop_p), RT.seed, off,
             ptr(bits), bits_ready, stream())
        ctx.save_for_backward(a, b_, c_, key_mask, bias, o, lse, bits)
        ctx.cfg = (mode, nh, float(drop_p), RT.seed, off, impl, scale)
        ctx.kv_slot = getattr(b_, "_kv_grad_slot", None) if mode == "cross" else None     # see hoisted_kv
        ctx.bias_slot = getattr(bias, "_dbias_slot", None) if bias is not None else None    # see graph_bias
        return o

    @staticmethod
    def backward(ctx, do):
        a, b_, c_, key_mask, bias, o, lse, bits = ctx.saved_tensors
        mode, nh, drop_p, seed, off, impl, scale = ctx.cfg
        do = do.contiguous()
        if mode == "self":
            H = a.shape[-1] // 3
            q, k, v = a[..., :H], a[..., H:2 * H], a[..., 2 * H:]
            da = torch.empty_like(a)
            dq, dk, dv = da[..., :H], da[..., H:2 * H], da[..., 2 * H:]
            grads = (da, None, None)
        elif mode == "cross":
            H = a.shape[-1]
            q, k, v = a, b_[..., :H], b_[..., H:]
            dq = torch.empty_like(a)
            # K/V projected for all layers of an encoder at once (hoisted_kv): the gradient goes straight into this
            # layer's column slice of the shared (B, Lk, layers * 2H) buffer, which feeds ONE input-gradient GEMM
            dkv = ctx.kv_slot[0].grad_slice(ctx.kv_slot[1], b_) if ctx.kv_slot is not None else torch.empty_like(b_)
            dk, dv = dkv[..., :H], dkv[..., H:]
            grads = (dq, dkv, None)
        else:
            q, k, v = a, b_, c_
            dq, dk, dv = torch.empty_like(a), torch.empty_like(b_), torch.empty_like(c_)
            grads = (dq, dk, dv)
        B, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]
        delta = torch.empty(B, nh, Lq, dtype=torch.float32, device=q.device)
        # the bias is shared by the heads: the kernels store per-head gradients (no atomics), summed here in a fixed order
        slot = ctx.bias_slot
        if bias is not None and ctx.needs_input_grad[5] and slot is not None and Lq == Lk:
            dbias_h = slot[0].slot(slot[1], B, Lq, q.device)       # shared (layers, B, nh, G, G) buffer, zeroed once
        else:
            slot = None
            dbias_h = torch.zeros(B, nh, Lq, Lk, dtype=torch.float32, device=q.device) \
                if (bias is not None and ctx.needs_input_grad[5]) else None
        assert do.shape == o.shape
        call("bevbert_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk),
             ptr(dv), ptr(dbias_h), ptr(key_mask), ptr(bias), _strides(q, k, v, o), B, nh, Lq, Lk, HEAD_DIM, scale,
             dtype_code(q), impl, drop_p, seed, off, ptr(bits), stream())
        if slot is not None:
            dbias = slot[0].dummy            # the real gradient sits in the holder; _GraphBias.backward reduces it
        else:
            dbias = None if dbias_h is None else dbias_h.sum(1)
        return (None,) + grads + (None, dbias, None, None, None)


class _BiasGradHolder:
    """Per-head gradients of one graph bias used by several attention layers: (layers, B, nh, G, G) fp32, zeroed once (the
    kernels write the valid entries only), filled slot by slot by the attention backward of each layer, reduced to the two
    scalars of sprel_linear by ONE launch (_GraphBias.backward) -- instead of, per layer, a zero fill, a head sum and
    autograd's addition of the layers' gradients, followed by three reductions."""

    def __init__(self, layers, nh):
        self.layers, self.nh, self.buf, self.dummy = layers, nh, None, None

    def slot(self, i, B, G, device):
        if self.buf is None:
            self.buf = torch.empty(self.layers, B, self.nh, G, G, dtype=torch.float32, device=device)
            call("bevbert_zero", ptr(self.buf), self.buf.numel() * 4, stream())
            self.dummy = torch.empty((), dtype=torch.float32, device=device).expand(B, G, G)    # never read: a shape for autograd
        return self.buf[i]


class _GraphBias(torch.autograd.Function):
    """bias = dists * w + b (vilmodel.py:575-577 sprel_linear), handed out once per attention layer."""

    @staticmethod
    def forward(ctx, dists, weight, bias, layers, nh):
        dists = dists.to(torch.float32).contiguous()
        out = torch.empty_like(dists)
        call("bevbert_graph_bias_fwd", ptr(dists), ptr(weight), ptr(bias), ptr(out), dists.numel(), stream())
        ctx.save_for_backward(dists)
        ctx.params = (weight, bias)
        ctx.holder = _BiasGradHolder(layers, nh)
        return tuple(out.view_as(out) for _ in range(layers))

    @staticmethod
    def backward(ctx, *grads):
        (dists,) = ctx.saved_tensors
        weight, bias = ctx.params
        h = ctx.holder
        if h.buf is None:
            return None, None, None, None, None
        sinks = []
        for p in (weight, bias):
            s = _sink(p) if p.requires_grad else None
            if s is not None:
                _mark_touched(p)
            sinks.append(s)
        B, G = dists.shape[0], dists.shape[-1]
        ws = torch.empty(1024, dtype=torch.float32, device=dists.device)
        call("bevbert_graph_bias_bwd", ptr(h.buf), ptr(dists), h.layers, B, h.nh, G, ptr(sinks[0]), ptr(sinks[1]), ptr(ws), stream())
        return None, None, None, None, None


def graph_bias(dists, weight, bias, layers, nh):
    """Per-layer views of the graph bias dists * w + b; each carries the slot its attention backward writes to."""
    outs = _GraphBias.apply(dists, weight, bias, layers, nh)
    holder = outs[0].grad_fn.holder if hasattr(outs[0].grad_fn, "holder") else None
    if holder is not None:
        for i, o in enumerate(outs):
            o._dbias_slot = (holder, i)
    return outs


def attention_self(qkv, key_mask, bias, nh, drop_p=0.0, training=False):
    return _Attention.apply("self", qkv, None, None, key_mask, bias, nh, drop_p if training else 0.0, RT.attn_impl)


def attention_cross(q, kv, key_mask, nh, drop_p=0.0, training=False):
    return _Attention.apply("cross", q, kv, None, key_mask, None, nh, drop_p if training else 0.0, RT.attn_impl)


def attention(q, k, v, key_mask=None, bias=None, nh=12, drop_p=0.0, training=False, impl=None):
    return _Attention.apply("sep", q, k, v, key_mask, bias, nh, drop_p if training else 0.0,
                            RT.attn_impl if impl is None else impl)


def attn_drop_bits(B, nh, Lq, Lk, drop_p, seed, offset, device):
    """Keep-bit workspace of one attention call ([forward layout | backward layout], int64 words); see
    include/bevbert_hip.h bevbert_attn_drop_bits."""
    bits = torch.empty(_drop_bits_words(B, nh, Lq, Lk), dtype=torch.int64, device=device)
    call("bevbert_attn_drop_bits", ptr(bits), B, nh, Lq, Lk, float(drop_p), int(seed), int(offset), stream())
    return bits
