"""Host-side mirror of the reference's cross-modal transformer (pretrain_src/model/vilmodel.py).

Class names, constructor arguments, forward signatures and -- exactly -- the ``state_dict`` keys follow the
reference, so its checkpoints load and its callers (pretrain_src/train_r2r.py, map_nav_src/r2r/agent.py) drop in.
The bodies do not: ``nn.Linear`` / ``nn.LayerNorm`` / ``nn.Embedding`` are used only as parameter holders, and every
forward is a short chain of fused HIP kernels (ops.py) and library GEMMs:

    BertAttention        = packed-QKV GEMM -> fused attention kernel -> GEMM -> fused bias+dropout+residual+LN
    BertIntermediate/Out = GEMM -> fused bias+GELU -> GEMM -> fused bias+dropout+residual+LN
    (N,12,Lq,Lk) scores, transpose_for_scores copies, (N,1,1,L) mask broadcasts: never materialised.

Host/device syncs of the reference forward (4608 item() + 192 nonzero at B=64) are gone: sequence
masks come from host-known shapes, the gmap aggregation is one CSR gather, the SAP fusion is one index gather.
"""

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import lib, ops
from .arena import ParamArena
from .config import BevBertConfig

LN_EPS = 1e-12
# captured steps (a side stream exists): the text-independent front of the map / BEV branches runs beside the text encoder
EARLY_BEV = __import__("os").environ.get("BEVBERT_EARLY_BEV", "0") == "1"     # measured SLOWER (round 4: 18.39 vs 18.23 ms; round 6: 18.45 vs 17.98 - 18.31): off


def gen_seq_masks(seq_lens, max_len):
    """ops.py:36-44 with a host-known max_len (the reference calls max() on a device tensor -> sync).  A loader that knows
    the lengths on the host ships the mask next to them (static_step.StaticBatch hangs it on the length tensor as
    ``_seq_masks``, with the additive fp32 forms the attention kernels read on the mask itself): then no arange / compare /
    cast / multiply launches are left inside the step for it."""
    pre = getattr(seq_lens, "_seq_masks", None)
    if pre is not None and pre.shape[1] == max_len:
        return pre
    return torch.arange(max_len, device=seq_lens.device)[None, :] < seq_lens[:, None]


def neg_key_mask(masks, value=-10000.0):
    """ops.py:25-34 extend_neg_masks, kept as (N, L) fp32 -- the kernels broadcast over heads and queries."""
    if masks is None:
        return None
    pre = getattr(masks, "_km", None)
    if pre is not None and value == -10000.0:
        return pre
    return ((1.0 - masks.to(torch.float32)) * value).contiguous()


class _Finalizable(nn.Module):
    """Modules that cache packed arena views implement _after_arena(arena, prefix)."""

    def _after_arena(self, arena, prefix):
        pass


def from_pretrained(cls, pretrained_model_name_or_path, config, state_dict):
    """transformers' ``PreTrainedModel.from_pretrained(None, config=..., state_dict=...)`` as both entry scripts call it
    (pretrain_src/train_r2r.py:153-155, map_nav_src/models/vlnbert_init.py:78-81): build from ``config`` (BERT
    initialisation), overlay ``state_dict`` non-strictly, tie weights, eval mode.  A shape mismatch raises (as there);
    keys the checkpoint lacks / has in excess are kept on ``model.load_report`` (transformers logs them)."""
    if pretrained_model_name_or_path is not None:
        raise NotImplementedError("checkpoints are passed as state_dict= (the entry scripts remap them first)")
    model = cls(config)
    missing, unexpected = [], []
    if state_dict:
        own = model.state_dict()
        bad = [k for k, v in state_dict.items() if k in own and tuple(v.shape) != tuple(own[k].shape)]
        if bad:
            raise RuntimeError("size mismatch for " + ", ".join(bad[:8]))
        res = model.load_state_dict(state_dict, strict=False)
        missing, unexpected = list(res.missing_keys), list(res.unexpected_keys)
    if hasattr(model, "tie_weights"):
        model.tie_weights()
    model.load_report = {"missing_keys": missing, "unexpected_keys": unexpected}
    model.eval()
    return model


def _p_of(child):
    """The probability of a dropout site is held by an ``nn.Dropout`` child (never called: the mask is drawn inside the
    fused HIP kernels) so that the reference's ``set_dropout(model, p)`` (pretrain_src/utils/misc.py:19-25), which
    rewrites ``module.p`` of every ``nn.Dropout`` it finds, reaches the kernels without a change to the loop."""
    return property(lambda self: getattr(self, child).p, lambda self, v: setattr(getattr(self, child), "p", v))


class BertEmbeddings(nn.Module):
    dropout_p = _p_of("dropout")

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.eps = config.layer_norm_eps

    def forward(self, input_ids, token_type_ids=None, position_ids=None):
        y = ops.embed_sum_layernorm(input_ids, self.word_embeddings.weight, self.position_embeddings.weight,
                                    self.token_type_embeddings.weight, self.LayerNorm.weight, self.LayerNorm.bias,
                                    self.eps, 0, padding_idx=self.word_embeddings.padding_idx)
        return ops.dropout(y, self.dropout_p, self.training)


class BertSelfAttention(_Finalizable):
    drop_p = _p_of("dropout")

    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        H = config.hidden_size
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def _after_arena(self, arena, prefix):
        H = self.query.weight.shape[0]
        wn = [f"{prefix}{k}.weight" for k in ("query", "key", "value")]
        bn = [f"{prefix}{k}.bias" for k in ("query", "key", "value")]
        wc, wg = arena.packed(wn, (3 * H, H))
        bc, bg = arena.packed(bn, (3 * H,))
        self.pw = ops._PackedParam([self.query.weight, self.key.weight, self.value.weight], wc, wg)
        self.pb = ops._PackedParam([self.query.bias, self.key.bias, self.value.bias], bc, bg)

    @staticmethod
    def arena_groups(prefix):
        return [[f"{prefix}{k}.weight" for k in ("query", "key", "value")],
                [f"{prefix}{k}.bias" for k in ("query", "key", "value")]]

    def forward(self, hidden_states, key_mask, bias=None):
        """Returns (attention output, hidden_states as residual tap): the caller adds the tap as its residual, and the
        residual's gradient is folded into the QKV projection's input-gradient GEMM (ops.linear_res)."""
        qkv, res = ops.linear_packed_res(hidden_states, self.pw, self.pb)
        return ops.attention_self(qkv, key_mask, bias, self.num_attention_heads, self.drop_p, self.training), res


class BertSelfOutput(nn.Module):
    drop_p = _p_of("dropout")

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.eps = config.layer_norm_eps

    def forward(self, hidden_states, input_tensor):
        h = ops.linear(hidden_states, self.dense.weight)            # bias folded into the fused kernel below
        return ops.bias_dropout_residual_layernorm(h, self.dense.bias, input_tensor, self.LayerNorm.weight,
                                                   self.LayerNorm.bias, self.eps, self.drop_p, self.training)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, key_mask, bias=None):
        a, res = self.self(input_tensor, key_mask, bias)
        return self.output(a, res)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        assert config.hidden_act == "gelu", "the path is specialised for erf-GELU (configs/*_model.json)"

    def forward(self, hidden_states):
        """Returns (gelu(dense(x)), x as residual tap) -- BertOutput adds the tap (vilmodel.py:168-193)."""
        h, res = ops.linear_res(hidden_states, self.dense.weight)
        return ops.bias_gelu(h, self.dense.bias), res


class BertOutput(nn.Module):
    drop_p = _p_of("dropout")

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.eps = config.layer_norm_eps

    def forward(self, hidden_states, input_tensor):
        h = ops.linear(hidden_states, self.dense.weight)
        return ops.bias_dropout_residual_layernorm(h, self.dense.bias, input_tensor, self.LayerNorm.weight,
                                                   self.LayerNorm.bias, self.eps, self.drop_p, self.training)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, key_mask):
        a = self.attention(hidden_states, key_mask)
        return self.output(*self.intermediate(a))


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.eps = config.layer_norm_eps

    def forward(self, hidden_states):
        h = ops.bias_gelu(ops.linear(hidden_states, self.dense.weight), self.dense.bias)
        return ops.layernorm(h, self.LayerNorm.weight, self.LayerNorm.bias, self.eps)


class BertLMPredictionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))

    def forward(self, hidden_states):
        h = self.transform(hidden_states)
        return ops.linear(h, self.decoder.weight, self.bias)


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


class BertOutAttention(_Finalizable):
    drop_p = _p_of("dropout")

    def __init__(self, config, ctx_dim=None):
        super().__init__()
        self.num_attention_heads = config.num_attention_heads
        H = config.hidden_size
        ctx_dim = H if ctx_dim is None else ctx_dim
        self.query = nn.Linear(H, H)
        self.key = nn.Linear(ctx_dim, H)
        self.value = nn.Linear(ctx_dim, H)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def _after_arena(self, arena, prefix):
        H, C = self.key.weight.shape
        wc, wg = arena.packed([f"{prefix}key.weight", f"{prefix}value.weight"], (2 * H, C))
        bc, bg = arena.packed([f"{prefix}key.bias", f"{prefix}value.bias"], (2 * H,))
        self.pw = ops._PackedParam([self.key.weight, self.value.weight], wc, wg)
        self.pb = ops._PackedParam([self.key.bias, self.value.bias], bc, bg)

    @staticmethod
    def arena_groups(prefix):
        return [[f"{prefix}key.weight", f"{prefix}value.weight"], [f"{prefix}key.bias", f"{prefix}value.bias"]]

    def forward(self, hidden_states, context, key_mask=None, kv=None):
        """Returns (attention output, hidden_states as residual tap).  ``kv``: this layer's (B, Lk, 2H) K|V projection of
        ``context`` when the encoder computed it for all its layers at once (CrossmodalEncoder.hoist_kv)."""
        q, res = ops.linear_res(hidden_states, self.query.weight, self.query.bias)
        if kv is None:
            kv = ops.linear_packed(context, self.pw, self.pb)
        return ops.attention_cross(q, kv, key_mask, self.num_attention_heads, self.drop_p, self.training), res


class BertXAttention(nn.Module):
    def __init__(self, config, ctx_dim=None):
        super().__init__()
        self.att = BertOutAttention(config, ctx_dim=ctx_dim)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, ctx_tensor, ctx_key_mask=None, ctx_kv=None):
        a, res = self.att(input_tensor, ctx_tensor, ctx_key_mask, ctx_kv)
        return self.output(a, res)


class GraphLXRTXLayer(nn.Module):
    """vilmodel.py:365-421; masks are (N, L) additive fp32 (or None), graph_sprels is (N, G, G) additive fp32."""

    def __init__(self, config):
        super().__init__()
        if config.use_lang2visn_attn:
            self.lang_self_att = BertAttention(config)
            self.lang_inter = BertIntermediate(config)
            self.lang_output = BertOutput(config)
        self.visn_self_att = BertAttention(config)
        self.visn_inter = BertIntermediate(config)
        self.visn_output = BertOutput(config)
        self.visual_attention = BertXAttention(config)

    def forward(self, lang_feats, lang_key_mask, visn_feats, visn_key_mask, graph_sprels=None, ctx_kv=None):
        a = self.visual_attention(visn_feats, lang_feats, lang_key_mask, ctx_kv)
        a = self.visn_self_att(a, visn_key_mask, graph_sprels)
        return self.visn_output(*self.visn_inter(a))

    def forward_lang2visn(self, lang_feats, lang_key_mask, visn_feats, visn_key_mask, ctx_kv=None):
        a = self.visual_attention(lang_feats, visn_feats, visn_key_mask, ctx_kv)
        a = self.lang_self_att(a, lang_key_mask)
        return self.lang_output(*self.lang_inter(a))

    def forward_visn2visn(self, visn_feats, visn_key_mask):
        a = self.visn_self_att(visn_feats, visn_key_mask)
        return self.visn_output(*self.visn_inter(a))


class LanguageEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.num_l_layers = config.num_l_layers
        self.update_lang_bert = config.update_lang_bert
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(self.num_l_layers)])
        if not self.update_lang_bert:
            for p in self.layer.parameters():
                p.requires_grad = False

    def forward(self, txt_embeds, txt_masks):
        km = neg_key_mask(txt_masks)
        for layer in self.layer:
            txt_embeds = layer(txt_embeds, km)
        if not self.update_lang_bert:
            txt_embeds = txt_embeds.detach()
        return txt_embeds


class CrossmodalEncoder(_Finalizable):
    def __init__(self, config):
        super().__init__()
        self.num_x_layers = config.num_x_layers
        self.x_layers = nn.ModuleList([GraphLXRTXLayer(config) for _ in range(self.num_x_layers)])
        self.kv_pw = self.kv_pb = None

    # The cross-attention context is the same tensor in every layer (vilmodel.py:383-398,446-463: lang_feats is never
    # updated; in the MLM direction the map / BEV tokens are not either), so the key / value projections of all layers
    # are ONE GEMM over weights laid out back to back in the arena (ops.hoisted_kv).
    @staticmethod
    def _kv_names(prefix, n_layers, kind):
        return [f"{prefix}x_layers.{i}.visual_attention.att.{k}.{kind}" for i in range(n_layers) for k in ("key", "value")]

    def arena_groups(self, prefix):
        return [self._kv_names(prefix, self.num_x_layers, "weight"), self._kv_names(prefix, self.num_x_layers, "bias")]

    def _after_arena(self, arena, prefix):
        n = self.num_x_layers
        att = self.x_layers[0].visual_attention.att
        H, C = att.key.weight.shape
        wc, wg = arena.packed(self._kv_names(prefix, n, "weight"), (n * 2 * H, C))
        bc, bg = arena.packed(self._kv_names(prefix, n, "bias"), (n * 2 * H,))
        ws = [p for l in self.x_layers for p in (l.visual_attention.att.key.weight, l.visual_attention.att.value.weight)]
        bs = [p for l in self.x_layers for p in (l.visual_attention.att.key.bias, l.visual_attention.att.value.bias)]
        self.kv_pw, self.kv_pb = ops._PackedParam(ws, wc, wg), ops._PackedParam(bs, bc, bg)

    def hoist_kv(self, context):
        """Per-layer (B, Lk, 2H) K|V views of ``context`` from one GEMM, or [None] * layers when hoisting is off."""
        if not ops.HOIST_KV or self.kv_pw is None or self.num_x_layers < 2:
            return [None] * self.num_x_layers
        return ops.hoisted_kv(context, self.kv_pw, self.kv_pb, self.num_x_layers)

    def packed_kv(self, context):
        """Inference only: the (B, Lk, layers * 2H) tensor ``hoist_kv`` slices, or None when hoisting is off.  A rollout
        computes it once per episode for the instruction (the text states are never updated in the map encoders,
        vilmodel.py:383-398) and hands it back through ``forward(..., kvs=split_kv(packed))`` at every navigation step."""
        if not ops.HOIST_KV or self.kv_pw is None or self.num_x_layers < 2:
            return None
        with torch.no_grad():
            return ops._linear_fwd(context, self.kv_pw.compute, self.kv_pb.compute)

    def split_kv(self, packed):
        w = packed.shape[-1] // self.num_x_layers
        return [packed[..., i * w:(i + 1) * w] for i in range(self.num_x_layers)]

    # Gradient exchange (train.PretrainTrainer, round 5): ``region_hook(k)`` is called when d loss / d (the streaming input
    # of x-layer k) is complete, i.e. when every backward kernel of the layers >= k of this encoder has been issued -- except
    # the hoisted K|V projection (one op for all layers: its backward runs after layer 0's) and the input embeddings.
    region_hook = None

    def _watch(self, k, x):
        if self.region_hook is not None and torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled():
            hook = self.region_hook
            hook(k, False)                                         # one more use of layer k in this forward
            x.register_hook(lambda g, k=k: (hook(k, True), g)[1])
        return x

    def forward(self, txt_embeds, txt_masks, img_embeds, img_masks, graph_sprels=None, kvs=None):
        tm, im = neg_key_mask(txt_masks), neg_key_mask(img_masks)
        if kvs is None:
            kvs = self.hoist_kv(txt_embeds)
        for k, (layer, kv) in enumerate(zip(self.x_layers, kvs)):
            sp = graph_sprels[k] if isinstance(graph_sprels, (tuple, list)) else graph_sprels      # per-layer views: ops.graph_bias
            img_embeds = layer(txt_embeds, tm, self._watch(k, img_embeds), im, graph_sprels=sp, ctx_kv=kv)
        return img_embeds

    def forward_lang2visn(self, txt_embeds, txt_key_mask, visn_feats, visn_key_mask, kvs=None):
        """The MLM direction (vilmodel.py:790-800): text queries over fixed map / BEV tokens in every layer.  ``kvs``: the
        hoisted K|V projections of ``visn_feats`` if the caller computed them ahead (they do not depend on the text)."""
        if kvs is None:
            kvs = self.hoist_kv(visn_feats)
        for k, (layer, kv) in enumerate(zip(self.x_layers, kvs)):
            txt_embeds = layer.forward_lang2visn(self._watch(k, txt_embeds), txt_key_mask, visn_feats, visn_key_mask, ctx_kv=kv)
        return txt_embeds


# ----------------------------------------------------------------------------- panorama encoder
class _MHAParams(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names (in_proj_weight, in_proj_bias, out_proj.*)."""

    def __init__(self, d_model):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)


class TransformerEncoderLayer(nn.Module):
    """transformer.py:133-182 with normalize_before=True (the only mode create_transformer_encoder uses)."""
    drop_p = _p_of("dropout")

    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = _MHAParams(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)      # eps 1e-5 (nn.LayerNorm default), transformer.py:144-145
        self.norm2 = nn.LayerNorm(d_model)
        self.nhead = nhead
        self.attn_drop_p = dropout              # nn.MultiheadAttention keeps a float: set_dropout does not reach it
        self.dropout = nn.Dropout(dropout)      # transformer.py:142,146-147 dropout / dropout1 / dropout2 share p

    def forward(self, src, key_mask, h, next_norm):
        """``h``: norm1(src) if the previous block already computed it (None: compute it here); ``next_norm``: (weight,
        bias, eps) of the LayerNorm that reads this layer's output (the next layer's norm1, or the encoder's final norm).
        Returns (src_out, next_norm(src_out)).
        The two residual adds of the block ride on the LayerNorm launches that follow them
        (ops.bias_dropout_residual_prenorm): 2 row-kernel launches per block instead of 4, and no separate dropout /
        gradient-add launches in backward."""
        tr, p = self.training, self.drop_p
        if h is None:
            h = ops.layernorm(src, self.norm1.weight, self.norm1.bias, 1e-5)
        qkv = ops.linear(h, self.self_attn.in_proj_weight, self.self_attn.in_proj_bias)
        a = ops.attention_self(qkv, key_mask, None, self.nhead, self.attn_drop_p, tr)
        o = ops.linear(a, self.self_attn.out_proj.weight)                   # bias folded into the fused kernel below
        h, src = ops.bias_dropout_residual_prenorm(o, self.self_attn.out_proj.bias, src, self.norm2.weight,
                                                   self.norm2.bias, 1e-5, p, tr)
        f = ops.bias_gelu(ops.linear(h, self.linear1.weight), self.linear1.bias)
        f = ops.linear(ops.dropout(f, p, tr), self.linear2.weight)
        w, b, eps = next_norm
        hn, src = ops.bias_dropout_residual_prenorm(f, self.linear2.bias, src, w, b, eps, p, tr)
        return src, hn


class TransformerEncoder(nn.Module):
    def __init__(self, config, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([
            TransformerEncoderLayer(config.hidden_size, config.num_attention_heads, config.intermediate_size,
                                    config.hidden_dropout_prob) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(config.hidden_size, eps=1e-12)      # ops.py:19-20

    def forward(self, src, src_key_padding_mask, key_mask=None, in_drop_p=None):
        """``key_mask``: the additive fp32 form of ``src_key_padding_mask`` (0 / -inf) if the caller has it already.
        ``in_drop_p``: the caller's dropout on ``src`` has NOT been applied yet -- it is drawn inside the launch of the first
        block's norm1 (the dropped tensor is the residual stream: without this its two consumers, norm1 and the first
        residual add, cost a dropout launch forward and a gradient-sum launch backward)."""
        km = key_mask
        if km is None and src_key_padding_mask is not None:      # boolean key_padding_mask -> -inf on padded keys (vilmodel.py:530-532)
            km = torch.zeros(src_key_padding_mask.shape, dtype=torch.float32, device=src.device)
            km = km.masked_fill(src_key_padding_mask, float("-inf")).contiguous()
        src = src.contiguous()
        h = None
        if in_drop_p:
            n1 = self.layers[0].norm1
            h, src = ops.bias_dropout_residual_prenorm(src, None, None, n1.weight, n1.bias, 1e-5, in_drop_p, True)
        for i, layer in enumerate(self.layers):
            nxt = self.layers[i + 1].norm1 if i + 1 < len(self.layers) else self.norm
            eps = 1e-5 if i + 1 < len(self.layers) else 1e-12
            src, h = layer(src, km, h, (nxt.weight, nxt.bias, eps))
        return h                     # = self.norm(src): computed by the last block's fused residual + LayerNorm


def _small_k_linear(x, lin, out_dtype):
    """Feature projections with K in {7, 10, 14}: run on the fp32 masters (K is not MFMA-tileable), emit the
    compute dtype.  The bias is NOT added here (the following fused LayerNorm kernel adds it)."""
    y = ops.linear(x.to(torch.float32), lin.weight, None, w_c=lin.weight)
    return y.to(out_dtype)


class ImageEmbeddings(nn.Module):
    drop_p = _p_of("dropout")

    def __init__(self, config):
        super().__init__()
        H = config.hidden_size
        self.img_linear = nn.Linear(config.image_feat_size, H)
        self.img_layer_norm = nn.LayerNorm(H, eps=1e-12)
        self.loc_linear = nn.Linear(getattr(config, "loc_feat_size", config.angle_feat_size + 3), H)
        self.loc_layer_norm = nn.LayerNorm(H, eps=1e-12)
        if getattr(config, "depth_feat_size", 0) > 0:      # bevbert_ce/pretrain/pretrain_src/model/vilmodel.py:473-477
            self.dep_linear = nn.Linear(config.depth_feat_size, H)
            self.dep_layer_norm = nn.LayerNorm(H, eps=1e-12)
        else:
            self.dep_linear = self.dep_layer_norm = None
        if config.obj_feat_size > 0 and config.obj_feat_size != config.image_feat_size:
            self.obj_linear = nn.Linear(config.obj_feat_size, H)
            self.obj_layer_norm = nn.LayerNorm(H, eps=1e-12)
        else:
            self.obj_linear = self.obj_layer_norm = None
        self.nav_type_embedding = nn.Embedding(getattr(config, "nav_type_vocab", 3), H)
        self.layer_norm = nn.LayerNorm(H, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.pano_encoder = TransformerEncoder(config, config.num_pano_layers) if config.num_pano_layers > 0 else None

    def embed(self, view_img_fts, loc_fts, nav_types, view_lens, type_embed_layer, obj_img_fts=None, obj_lens=None,
              view_dep_fts=None):
        """Shared by forward (pre-training) and forward_panorama_per_step (fine-tuning).

        Object tokens (REVERIE / SOON, vilmodel.py:502-516) follow the views of their panorama: the reference
        concatenates and re-pads per panorama in a Python loop; here it is one gather over
        [views | objects | zero row] with an index built on the device from the two length vectors."""
        cd = ops._compute(self.img_linear.weight).dtype
        x = ops.linear(view_img_fts.to(cd), self.img_linear.weight)
        e = ops.bias_dropout_residual_layernorm(x, self.img_linear.bias, None, self.img_layer_norm.weight,
                                                self.img_layer_norm.bias, 1e-12)
        if self.dep_linear is not None:         # CE fork (its vilmodel.py:507-509): + LN(dep_linear(depth features))
            assert view_dep_fts is not None, "this configuration (depth_feat_size > 0) needs traj_view_dep_fts"
            xd = ops.linear(view_dep_fts.to(cd), self.dep_linear.weight)
            e = e + ops.bias_dropout_residual_layernorm(xd, self.dep_linear.bias, None, self.dep_layer_norm.weight,
                                                        self.dep_layer_norm.bias, 1e-12)
        lens = view_lens
        if obj_img_fts is not None:
            lin, ln = (self.img_linear, self.img_layer_norm) if self.obj_linear is None \
                else (self.obj_linear, self.obj_layer_norm)
            xo = ops.linear(obj_img_fts.to(cd), lin.weight)
            eo = ops.bias_dropout_residual_layernorm(xo, lin.bias, None, ln.weight, ln.bias, 1e-12)
            V, O, L = e.shape[1], eo.shape[1], loc_fts.shape[1]
            lens = view_lens + obj_lens
            pos = torch.arange(L, device=e.device)[None, :]
            vl, tl = view_lens[:, None], lens[:, None]
            idx = torch.where(pos < vl, pos, torch.where(pos < tl, V + pos - vl, torch.full_like(pos, V + O)))
            pool = torch.cat([e, eo, e.new_zeros(e.shape[0], 1, e.shape[2])], 1)
            e = torch.gather(pool, 1, idx[..., None].expand(-1, -1, e.shape[2]))
        # ((e + LN(loc_linear(loc))) + nav_type) + token_type: the first two sums ride on the store of the LayerNorm; with
        # arena parameters the K = 7 projection itself is computed inside that kernel too (ops.smallk_linear_layernorm_plus)
        if ops.smallk_linear_layernorm_plus_supported(loc_fts, self.loc_linear, self.loc_layer_norm, e, self.nav_type_embedding):
            e = ops.smallk_linear_layernorm_plus(loc_fts, self.loc_linear, self.loc_layer_norm, 1e-12, e,
                                                 self.nav_type_embedding, nav_types)
        else:
            loc = _small_k_linear(loc_fts, self.loc_linear, cd)
            e = ops.bias_layernorm_plus(loc, self.loc_linear.bias, self.loc_layer_norm.weight, self.loc_layer_norm.bias, 1e-12,
                                        e, embedding_lookup(self.nav_type_embedding, nav_types))
        if getattr(type_embed_layer.weight, "main_grad", None) is not None or not type_embed_layer.weight.requires_grad:
            # + token-type row 1 as the broadcast bias of the final LayerNorm (added in fp32 inside the kernel; its gradient
            # is the kernel's deterministic column reduction: ops.RowOfTable)
            e = ops.bias_dropout_residual_layernorm(e, ops.RowOfTable(type_embed_layer.weight, 1), None,
                                                    self.layer_norm.weight, self.layer_norm.bias, 1e-12)
        else:       # plain-tensor parameters (no arena): the torch composition
            e = e + embedding_lookup(type_embed_layer, torch.ones(1, 1, dtype=torch.long, device=e.device))
            e = ops.layernorm(e, self.layer_norm.weight, self.layer_norm.bias, 1e-12)
        fuse_drop = self.pano_encoder is not None and self.training and self.drop_p > 0 and e.is_cuda
        if not fuse_drop:
            e = ops.dropout(e, self.drop_p, self.training)
        masks = gen_seq_masks(lens, e.shape[1])
        if self.pano_encoder is not None:
            km = getattr(masks, "_km_inf", None)          # loader-built (static_step.StaticBatch)
            e = self.pano_encoder(e, None if km is not None else masks.logical_not(), key_mask=km,
                                  in_drop_p=self.drop_p if fuse_drop else None)
        return e, masks

    def forward(self, traj_view_img_fts, traj_obj_img_fts, traj_loc_fts, traj_nav_types, traj_step_lens,
                traj_vp_view_lens, traj_vp_obj_lens, type_embed_layer, traj_view_dep_fts=None):
        e, _ = self.embed(traj_view_img_fts, traj_loc_fts, traj_nav_types, traj_vp_view_lens, type_embed_layer,
                          traj_obj_img_fts, traj_vp_obj_lens, traj_view_dep_fts)
        lens = traj_vp_view_lens if traj_obj_img_fts is None else traj_vp_view_lens + traj_vp_obj_lens
        if e.shape[0] != sum(traj_step_lens):
            # a static_step.StaticBatch appends dummy panoramas (PANO_PAD) that only the flat embed() / _traj() path may see
            raise ValueError(f"{e.shape[0]} panoramas for step lengths summing to {sum(traj_step_lens)}: per-trajectory "
                             "splits need the unpadded batch (reference API), not a StaticBatch")
        return torch.split(e, traj_step_lens, 0), torch.split(lens, traj_step_lens, 0)


class _GatherRows(torch.autograd.Function):
    """table[idx] on the compute copy; the gradient is index_add'ed into the master's fp32 gradient arena."""

    @staticmethod
    def forward(ctx, idx, table, table_c):
        ctx.save_for_backward(idx)
        ctx.table = table
        return table_c[idx]

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        table = ctx.table
        if not table.requires_grad:
            return None, None, None
        sink = ops._sink(table)
        d = dy.reshape(-1, dy.shape[-1])
        if sink is not None and d.is_cuda and d.shape[1] % 4 == 0 and ops.WgradStream.DEFER_FINALIZE:
            # the tables here have 2..100 rows and up to 28 224 gradient rows: per-(table row, row slice) sums without
            # atomics (index_add_ piles 0.3 ms of atomics onto two destination rows), folded into the arena by the
            # step's batched accumulate -- one launch here, none for the fold
            ops._mark_touched(table)
            ops.embedding_grad_small(idx.reshape(-1), d.contiguous(), sink, table.shape[0])
            return None, None, None
        # fallback (CPU reference runs, no arena): a one-hot GEMM (rows x N) @ (N x H), deterministic as well
        onehot = F.one_hot(idx.reshape(-1), table.shape[0]).to(d.dtype)
        g = onehot.t().mm(d)
        if sink is not None:
            ops._mark_touched(table)
            # sink writes live on the weight-gradient stream (the token-type table is also written by _EmbedLN there)
            ops.WgradStream.submit(dy.device, lambda: ops._on_launch_stream(lambda: sink.add_(g)), g, dy)
            return None, None, None
        return None, g.to(table.dtype), None


def embedding_lookup(emb: nn.Embedding, idx):
    return _GatherRows.apply(idx, emb.weight, ops._compute(emb.weight))


class LocalBEVEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.bev_dim = config.bev_dim
        H = config.hidden_size
        self.bev_fts_embeddings = nn.Sequential(nn.Linear(768, H), nn.LayerNorm(H, eps=1e-12))
        self.bev_pos_embeddings = nn.Sequential(nn.Linear(3 + 7, H), nn.LayerNorm(H, eps=1e-12))
        self.nav_type_embedding = nn.Embedding(2, H)
        self.encoder = CrossmodalEncoder(config)

    def bev_input_embedding(self, bev_fts, bev_pos_fts, bev_nav_masks):
        lin, ln = self.bev_fts_embeddings[0], self.bev_fts_embeddings[1]
        cd = ops._compute(lin.weight).dtype
        x = ops.linear(bev_fts.to(cd), lin.weight)
        e = ops.bias_dropout_residual_layernorm(x, lin.bias, None, ln.weight, ln.bias, 1e-12)
        lin, ln = self.bev_pos_embeddings[0], self.bev_pos_embeddings[1]
        nav_idx = getattr(bev_nav_masks, "_long", None)            # loader-built (static_step.StaticBatch)
        if nav_idx is None:
            nav_idx = bev_nav_masks.long()
        # (e + LN(pos)) + nav_type in the LayerNorm's own launch (and the K = 10 projection with it: smallk.hip)
        if ops.smallk_linear_layernorm_plus_supported(bev_pos_fts, lin, ln, e, self.nav_type_embedding):
            return ops.smallk_linear_layernorm_plus(bev_pos_fts, lin, ln, 1e-12, e, self.nav_type_embedding, nav_idx)
        pos = _small_k_linear(bev_pos_fts, lin, cd)
        return ops.bias_layernorm_plus(pos, lin.bias, ln.weight, ln.bias, 1e-12, e,
                                       embedding_lookup(self.nav_type_embedding, nav_idx))

    def with_objects(self, bev_embeds, bev_masks, obj_embeds, obj_masks):
        """vilmodel.py:601-606: object tokens are appended to the BEV cells (an all-ones BEV mask may come as None)."""
        if obj_embeds is None:
            return bev_embeds, bev_masks
        if bev_masks is None:
            bev_masks = torch.ones(bev_embeds.shape[:2], dtype=torch.bool, device=bev_embeds.device)
        return torch.cat([bev_embeds, obj_embeds], 1), torch.cat([bev_masks, obj_masks], 1)

    def forward(self, txt_embeds, txt_masks, bev_fts, bev_pos_fts, bev_masks, bev_nav_masks, obj_embeds, obj_masks,
                bev_in=None, txt_kvs=None):
        """``bev_in``: the input embedding if the caller has computed it already (on a side stream, beside the text
        encoder: it does not depend on the text).  ``txt_kvs``: CrossmodalEncoder.split_kv of a cached packed_kv."""
        bev_embeds = bev_in if bev_in is not None else self.bev_input_embedding(bev_fts, bev_pos_fts, bev_nav_masks)
        x, m = self.with_objects(bev_embeds, bev_masks, obj_embeds, obj_masks)
        x = self.encoder(txt_embeds, txt_masks, x.contiguous(), m, kvs=txt_kvs)
        K = self.bev_dim * self.bev_dim
        if obj_embeds is None:
            return x, None
        return x[:, :K], x[:, K:]


def gmap_csr_arrays(traj_step_lens, view_lens_host, traj_vpids, traj_cand_vpids, gmap_vpids, n_views, G=None):
    """vilmodel.py:632-666 (_aggregate_gmap_features) as a CSR over the flattened (sum_T * V) token rows: returns
    (rowptr, idx, w, n_src, G).  Output row b*G + j (G = batch max incl. [stop], or a larger padded width); [stop] and
    padding rows are empty segments (-> zeros)."""
    B = len(traj_step_lens)
    G = max(max(len(g) for g in gmap_vpids), G or 0)
    rowptr, idx, w = [0], [], []
    t0 = 0
    for i in range(B):
        T = traj_step_lens[i]
        visited, unvisited = {}, {}
        for t in range(T):
            visited[traj_vpids[i][t]] = t
            for j, vp in enumerate(traj_cand_vpids[i][t]):
                if vp not in visited:
                    unvisited.setdefault(vp, []).append((t, j))
        for j in range(G):
            if 0 < j < len(gmap_vpids[i]):
                vp = gmap_vpids[i][j]
                if vp in visited:
                    t = visited[vp]
                    n = int(view_lens_host[t0 + t])
                    base = (t0 + t) * n_views
                    idx.extend(range(base, base + n))
                    w.extend([1.0 / n] * n)
                else:
                    toks = unvisited[vp]
                    idx.extend((t0 + t) * n_views + jj for t, jj in toks)
                    w.extend([1.0 / len(toks)] * len(toks))
            rowptr.append(len(idx))
        t0 += T
    return rowptr, idx, w, t0 * n_views, G


def build_gmap_csr(traj_step_lens, view_lens_host, traj_vpids, traj_cand_vpids, gmap_vpids, n_views, device, G=None,
                   capacity=None):
    rowptr, idx, w, n_src, G = gmap_csr_arrays(traj_step_lens, view_lens_host, traj_vpids, traj_cand_vpids, gmap_vpids,
                                               n_views, G)
    return ops.SegmentCSR(rowptr, idx, w, n_src, device, capacity=capacity), G


class GlobalMapEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        H = config.hidden_size
        self.gmap_pos_embeddings = nn.Sequential(nn.Linear(config.angle_feat_size + 3, H), nn.LayerNorm(H, eps=1e-12))
        self.gmap_step_embeddings = nn.Embedding(config.max_action_steps, H)
        self.encoder = CrossmodalEncoder(config)
        self.sprel_linear = nn.Linear(1, 1) if config.graph_sprels else None

    def pos_step_embedding(self, gmap_img_fts, gmap_step_ids, gmap_pos_fts):
        lin, ln = self.gmap_pos_embeddings[0], self.gmap_pos_embeddings[1]
        if ops.smallk_linear_layernorm_plus_supported(gmap_pos_fts, lin, ln, gmap_img_fts, self.gmap_step_embeddings):
            return ops.smallk_linear_layernorm_plus(gmap_pos_fts, lin, ln, 1e-12, gmap_img_fts, self.gmap_step_embeddings,
                                                    gmap_step_ids)
        pos = _small_k_linear(gmap_pos_fts, lin, gmap_img_fts.dtype)
        # (LN(pos) + img) + step embedding on the store of the LayerNorm: one rounding of the sum in bf16 instead of three
        # (vilmodel.py:589-593 adds the same three terms)
        return ops.bias_layernorm_plus(pos, lin.bias, ln.weight, ln.bias, 1e-12, gmap_img_fts.contiguous(),
                                       embedding_lookup(self.gmap_step_embeddings, gmap_step_ids))

    def gmap_input_embedding(self, traj_embeds_flat, csr, G, gmap_step_ids, gmap_pos_fts, gmap_lens):
        B = gmap_step_ids.shape[0]
        img = ops.segment_wsum(traj_embeds_flat, csr).view(B, G, -1)
        return self.pos_step_embedding(img, gmap_step_ids, gmap_pos_fts), gen_seq_masks(gmap_lens, G)

    def sprels(self, gmap_pair_dists):
        """(B, G, G) fp32 additive bias; with arena parameters on the GPU a tuple with one view per x-layer (ops.graph_bias:
        one launch forward, one launch for the gradients of all layers)."""
        if self.sprel_linear is None:
            return None
        lw, lb = self.sprel_linear.weight, self.sprel_linear.bias
        if gmap_pair_dists.is_cuda and all(p.dtype == torch.float32 and (not p.requires_grad or ops._sink(p) is not None) for p in (lw, lb)):
            n_layers = self.encoder.num_x_layers
            nh = self.encoder.x_layers[0].visn_self_att.self.num_attention_heads
            return ops.graph_bias(gmap_pair_dists, lw, lb, n_layers, nh)
        w, b = ops.use_param(lw).view(()), ops.use_param(lb).view(())
        return (gmap_pair_dists.float() * w + b).contiguous()

    def forward(self, txt_embeds, txt_masks, gmap_embeds, gmap_masks, gmap_pair_dists, txt_kvs=None):
        return self.encoder(txt_embeds, txt_masks, gmap_embeds, gmap_masks, graph_sprels=self.sprels(gmap_pair_dists),
                            kvs=txt_kvs)


def _host_list(x):
    return x.tolist() if torch.is_tensor(x) else list(x)


class GlocalTextPathCMT(nn.Module):
    """vilmodel.py:703-883.  Same 20-argument positional signature for forward / forward_mlm / forward_sem."""

    def __init__(self, config):
        super().__init__()
        config = BevBertConfig.adopt(config)
        self.config = config
        self.bev_dim = config.bev_dim
        self.embeddings = BertEmbeddings(config)
        self.lang_encoder = LanguageEncoder(config)
        self.img_embeddings = ImageEmbeddings(config)
        self.local_encoder = LocalBEVEncoder(config)
        self.global_encoder = GlobalMapEncoder(config)

    # -- shared stages -------------------------------------------------------------------------
    def _text(self, txt_ids, txt_lens):
        txt_masks = gen_seq_masks(txt_lens, txt_ids.shape[1])
        return self.lang_encoder(self.embeddings(txt_ids), txt_masks), txt_masks

    def _traj(self, traj_view_img_fts, traj_loc_fts, traj_nav_types, traj_vp_view_lens, traj_obj_img_fts=None,
              traj_vp_obj_lens=None, traj_view_dep_fts=None):
        e, masks = self.img_embeddings.embed(traj_view_img_fts, traj_loc_fts, traj_nav_types, traj_vp_view_lens,
                                             self.embeddings.token_type_embeddings, traj_obj_img_fts, traj_vp_obj_lens,
                                             traj_view_dep_fts)
        return e

    @staticmethod
    def _token_lens_host(traj_vp_view_lens, traj_vp_obj_lens, view_lens_host, obj_lens_host):
        """Host copy of the per-panorama token counts (views + objects): from the loader's CPU copies when given."""
        vl = view_lens_host if view_lens_host is not None else traj_vp_view_lens
        vl = _host_list(vl)
        if traj_vp_obj_lens is None:
            return vl, None
        ol = _host_list(obj_lens_host if obj_lens_host is not None else traj_vp_obj_lens)
        return [a + b for a, b in zip(vl, ol)], ol

    def _obj_tokens(self, traj, traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens, obj_lens_host):
        """vilmodel.py:748-756: the object tokens of every sample's LAST panorama, zero padded, + validity mask."""
        if traj_vp_obj_lens is None:
            return None, None
        ends_host = np.cumsum(traj_step_lens) - 1
        O = max(1, max(obj_lens_host[e] for e in ends_host))
        ends = torch.from_numpy(ends_host).to(traj.device, non_blocking=True)
        vl, ol = traj_vp_view_lens[ends], traj_vp_obj_lens[ends]
        j = torch.arange(O, device=traj.device)[None, :]
        valid = j < ol[:, None]
        idx = torch.where(valid, vl[:, None] + j, torch.zeros_like(j))
        rows = traj[ends]                                           # (B, L, H)
        obj = torch.gather(rows, 1, idx[..., None].expand(-1, -1, rows.shape[2]))
        return obj * valid[..., None].to(obj.dtype), valid

    def _gmap_inputs(self, traj_embeds, traj_step_lens, traj_vp_view_lens, traj_vpids, traj_cand_vpids, gmap_vpids,
                     gmap_step_ids, gmap_pos_fts, gmap_lens, view_lens_host=None, gmap_csr=None):
        V = traj_embeds.shape[1]
        if gmap_csr is not None:        # built by the loader (synthetic.batch_to) next to the host->device copies
            csr, G = gmap_csr
            assert csr.n_src == traj_embeds.shape[0] * V and csr.n_out == len(traj_step_lens) * G
        else:
            if view_lens_host is None:
                view_lens_host = _host_list(traj_vp_view_lens)        # one small D2H copy when no host copy is given
            csr, G = build_gmap_csr(traj_step_lens, view_lens_host, traj_vpids, traj_cand_vpids, gmap_vpids, V,
                                    traj_embeds.device)
        assert G == gmap_step_ids.shape[1]
        flat = traj_embeds.reshape(-1, traj_embeds.shape[-1])
        return self.global_encoder.gmap_input_embedding(flat, csr, G, gmap_step_ids, gmap_pos_fts, gmap_lens)

    # -- reference entry points ----------------------------------------------------------------
    def forward(self, txt_ids, txt_lens, traj_view_img_fts, traj_obj_img_fts, traj_loc_fts, traj_nav_types,
                traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens, traj_vpids, traj_cand_vpids,
                gmap_lens, gmap_step_ids, gmap_pos_fts, gmap_pair_dists, gmap_vpids,
                bev_fts, bev_pos_fts, bev_masks, bev_nav_masks, return_gmap_embeds=True, view_lens_host=None,
                obj_lens_host=None, gmap_csr=None, traj_view_dep_fts=None):
        has_obj = traj_obj_img_fts is not None
        br = ops.Branches(txt_ids.device)
        gmap_embeds = obj_embeds = obj_masks = traj = None
        need_traj = return_gmap_embeds or has_obj
        # side stream: panorama encoder (independent of the text);  current stream: text encoder
        br.fork(traj_view_img_fts, traj_loc_fts, traj_nav_types, traj_vp_view_lens, traj_obj_img_fts, traj_vp_obj_lens,
                traj_view_dep_fts)
        if need_traj:
            with br.side():
                traj = self._traj(traj_view_img_fts, traj_loc_fts, traj_nav_types, traj_vp_view_lens, traj_obj_img_fts,
                                  traj_vp_obj_lens, traj_view_dep_fts)
            tok_lens, ol_host = self._token_lens_host(traj_vp_view_lens, traj_vp_obj_lens, view_lens_host,
                                                      obj_lens_host)
        # ... and, behind it, the BEV input embedding (28 224-row GEMM + two LayerNorms): it does not depend on the text
        # either, and the text encoder's 5 120-row kernels leave most of the chip idle (round-4 timeline: 1.3 busy queues
        # on average during the first 4.3 ms of a step)
        bev_in = None
        if br.on and EARLY_BEV and not has_obj:
            br.fork(bev_fts, bev_pos_fts, bev_nav_masks)
            with br.side():
                bev_in = self.local_encoder.bev_input_embedding(bev_fts, bev_pos_fts, bev_nav_masks)
        txt_embeds, txt_masks = self._text(txt_ids, txt_lens)
        if bev_in is not None:
            br.join(bev_in)
        if has_obj:             # the object tokens feed the BEV branch: they are needed on the current stream
            br.join(traj)
            obj_embeds, obj_masks = self._obj_tokens(traj, traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens,
                                                     ol_host)
        # side stream: global-map encoder (tiny kernels);  current stream: BEV encoder (28 224-row kernels)
        if return_gmap_embeds:
            br.fork(txt_embeds, txt_masks, gmap_step_ids, gmap_pos_fts, gmap_lens, gmap_pair_dists, traj)
            with br.side():
                g_in, g_masks = self._gmap_inputs(traj, traj_step_lens, traj_vp_view_lens, traj_vpids,
                                                  traj_cand_vpids, gmap_vpids, gmap_step_ids, gmap_pos_fts, gmap_lens,
                                                  tok_lens, gmap_csr)
                gmap_embeds = self.global_encoder(txt_embeds, txt_masks, g_in, g_masks, gmap_pair_dists)
        bev_embeds, obj_embeds = self.local_encoder(txt_embeds, txt_masks, bev_fts, bev_pos_fts,
                                                    _all_ones_to_none(bev_masks), bev_nav_masks, obj_embeds, obj_masks,
                                                    bev_in=bev_in)
        br.join(gmap_embeds)
        return gmap_embeds, bev_embeds, obj_embeds, obj_masks

    def forward_mlm(self, txt_ids, txt_lens, traj_view_img_fts, traj_obj_img_fts, traj_loc_fts, traj_nav_types,
                    traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens, traj_vpids, traj_cand_vpids,
                    gmap_lens, gmap_step_ids, gmap_pos_fts, gmap_pair_dists, gmap_vpids,
                    bev_fts, bev_pos_fts, bev_masks, bev_nav_masks, view_lens_host=None, obj_lens_host=None,
                    gmap_csr=None, traj_view_dep_fts=None):
        br = ops.Branches(txt_ids.device)
        br.fork(traj_view_img_fts, traj_loc_fts, traj_nav_types, traj_vp_view_lens, traj_obj_img_fts, traj_vp_obj_lens,
                traj_view_dep_fts, bev_fts, bev_pos_fts, bev_nav_masks, gmap_step_ids, gmap_pos_fts, gmap_lens)
        tok_lens, ol_host = self._token_lens_host(traj_vp_view_lens, traj_vp_obj_lens, view_lens_host, obj_lens_host)
        early = br.on and EARLY_BEV and traj_obj_img_fts is None
        g_in = g_masks = g_kvs = bev_in = bev_kvs = None
        with br.side():         # panorama encoder next to the text encoder
            traj = self._traj(traj_view_img_fts, traj_loc_fts, traj_nav_types, traj_vp_view_lens, traj_obj_img_fts,
                              traj_vp_obj_lens, traj_view_dep_fts)
            if early:
                # ... and everything else of the step that does not depend on the text: the map / BEV input embeddings and
                # the K|V projections of all cross-attention layers over them (one 28 224 x 6 144 x 768 GEMM for the BEV):
                # ~0.5 ms of large kernels beside the text encoder's small ones instead of behind them
                g_in, g_masks = self._gmap_inputs(traj, traj_step_lens, traj_vp_view_lens, traj_vpids, traj_cand_vpids,
                                                  gmap_vpids, gmap_step_ids, gmap_pos_fts, gmap_lens, tok_lens, gmap_csr)
                g_kvs = self.global_encoder.encoder.hoist_kv(g_in)
                bev_in = self.local_encoder.bev_input_embedding(bev_fts, bev_pos_fts, bev_nav_masks).contiguous()
                bev_kvs = self.local_encoder.encoder.hoist_kv(bev_in)
        txt_embeds, txt_masks = self._text(txt_ids, txt_lens)
        tm = neg_key_mask(txt_masks)
        if early:
            br.join(bev_in, *[t for t in bev_kvs if t is not None])
        # side stream: text queries over the global map;  current stream: text queries over the BEV (+ objects)
        br.fork(txt_embeds, tm, gmap_step_ids, gmap_pos_fts, gmap_lens)
        with br.side():
            if not early:
                g_in, g_masks = self._gmap_inputs(traj, traj_step_lens, traj_vp_view_lens, traj_vpids, traj_cand_vpids,
                                                  gmap_vpids, gmap_step_ids, gmap_pos_fts, gmap_lens, tok_lens, gmap_csr)
            gm = neg_key_mask(g_masks)
            g_txt = self.global_encoder.encoder.forward_lang2visn(txt_embeds, tm, g_in, gm, kvs=g_kvs)
        if not early:
            bev_in = self.local_encoder.bev_input_embedding(bev_fts, bev_pos_fts, bev_nav_masks)
        obj_embeds = obj_masks = None
        if traj_obj_img_fts is not None:
            br.join(traj)
            obj_embeds, obj_masks = self._obj_tokens(traj, traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens, ol_host)
        bev_in, bev_obj_masks = self.local_encoder.with_objects(bev_in, _all_ones_to_none(bev_masks), obj_embeds,
                                                                obj_masks)
        bev_in = bev_in.contiguous()
        bm = neg_key_mask(bev_obj_masks)
        b_txt = self.local_encoder.encoder.forward_lang2visn(txt_embeds, tm, bev_in, bm, kvs=bev_kvs)
        br.join(g_txt)
        return g_txt + b_txt

    def forward_sem(self, txt_ids, txt_lens, traj_view_img_fts, traj_obj_img_fts, traj_loc_fts, traj_nav_types,
                    traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens, traj_vpids, traj_cand_vpids,
                    gmap_lens, gmap_step_ids, gmap_pos_fts, gmap_pair_dists, gmap_vpids,
                    bev_fts, bev_pos_fts, bev_masks, bev_nav_masks, sem_pred_token=None, view_lens_host=None,
                    obj_lens_host=None, gmap_csr=None, traj_view_dep_fts=None):
        bm = _all_ones_to_none(bev_masks)
        if sem_pred_token == "cattn":
            txt_embeds, txt_masks = self._text(txt_ids, txt_lens)
            obj_embeds = obj_masks = None
            if traj_obj_img_fts is not None:
                traj = self._traj(traj_view_img_fts, traj_loc_fts, traj_nav_types, traj_vp_view_lens,
                                  traj_obj_img_fts, traj_vp_obj_lens, traj_view_dep_fts)
                _, ol_host = self._token_lens_host(traj_vp_view_lens, traj_vp_obj_lens, view_lens_host, obj_lens_host)
                obj_embeds, obj_masks = self._obj_tokens(traj, traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens,
                                                         ol_host)
            # (without objects the reference still runs img_embeddings here, vilmodel.py:847-851, but nothing consumes it)
            bev_embeds, _ = self.local_encoder(txt_embeds, txt_masks, bev_fts, bev_pos_fts, bm, bev_nav_masks,
                                               obj_embeds, obj_masks)
        elif sem_pred_token == "sattn":
            bev_embeds = self.local_encoder.bev_input_embedding(bev_fts, bev_pos_fts, bev_nav_masks)
            km = neg_key_mask(bm)
            for k, layer in enumerate(self.local_encoder.encoder.x_layers):
                bev_embeds = layer.forward_visn2visn(self.local_encoder.encoder._watch(k, bev_embeds), km)
        elif sem_pred_token == "embed":
            bev_embeds = self.local_encoder.bev_input_embedding(bev_fts, bev_pos_fts, bev_nav_masks)
        else:
            raise NotImplementedError
        return bev_embeds


def _all_ones_to_none(mask):
    """bev_masks is forced to all-ones upstream (pretrain_cmt.py:152, agent.py:187); a Python ``True`` marker
    (set by our lift_splat) lets the kernels skip the mask entirely.  A real tensor mask is honoured as is."""
    if mask is None or mask is True:
        return None
    return mask


# ----------------------------------------------------------------------------- arena wiring
class GlocalTextPathCMTCE(GlocalTextPathCMT):
    """The continuous-environment fork's positional signatures (bevbert_ce/pretrain/pretrain_src/model/vilmodel.py:717-
    776,778-840): ``traj_view_dep_fts`` follows ``traj_view_img_fts`` and ``forward`` returns (gmap_embeds, bev_embeds)."""

    def forward(self, txt_ids, txt_lens, traj_view_img_fts, traj_view_dep_fts, traj_obj_img_fts, traj_loc_fts,
                traj_nav_types, traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens, traj_vpids, traj_cand_vpids,
                gmap_lens, gmap_step_ids, gmap_pos_fts, gmap_pair_dists, gmap_vpids,
                bev_fts, bev_pos_fts, bev_masks, bev_nav_masks, return_gmap_embeds=True, **host_kw):
        g, b, _, _ = super().forward(txt_ids, txt_lens, traj_view_img_fts, traj_obj_img_fts, traj_loc_fts,
                                     traj_nav_types, traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens, traj_vpids,
                                     traj_cand_vpids, gmap_lens, gmap_step_ids, gmap_pos_fts, gmap_pair_dists,
                                     gmap_vpids, bev_fts, bev_pos_fts, bev_masks, bev_nav_masks,
                                     return_gmap_embeds=return_gmap_embeds, traj_view_dep_fts=traj_view_dep_fts,
                                     **host_kw)
        return g, b

    def forward_mlm(self, txt_ids, txt_lens, traj_view_img_fts, traj_view_dep_fts, traj_obj_img_fts, traj_loc_fts,
                    traj_nav_types, traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens, traj_vpids, traj_cand_vpids,
                    gmap_lens, gmap_step_ids, gmap_pos_fts, gmap_pair_dists, gmap_vpids,
                    bev_fts, bev_pos_fts, bev_masks, bev_nav_masks, **host_kw):
        return super().forward_mlm(txt_ids, txt_lens, traj_view_img_fts, traj_obj_img_fts, traj_loc_fts,
                                   traj_nav_types, traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens, traj_vpids,
                                   traj_cand_vpids, gmap_lens, gmap_step_ids, gmap_pos_fts, gmap_pair_dists,
                                   gmap_vpids, bev_fts, bev_pos_fts, bev_masks, bev_nav_masks,
                                   traj_view_dep_fts=traj_view_dep_fts, **host_kw)


def arena_groups(module):
    """Parameter runs that must be contiguous in the arena: packed QKV per self-attention, K|V per cross-attention --
    and, where a CrossmodalEncoder hoists them, the K|V of all its layers back to back (the per-layer pairs stay
    adjacent inside that run, so BertOutAttention's own packed views remain valid)."""
    groups, taken = [], set()
    for name, m in module.named_modules():
        if isinstance(m, CrossmodalEncoder) and m.num_x_layers >= 2:
            for g in m.arena_groups(name + "." if name else ""):
                groups.append(g)
                taken.update(g)
    for name, m in module.named_modules():
        if isinstance(m, (BertSelfAttention, BertOutAttention)):
            for g in m.arena_groups(name + "." if name else ""):
                if not taken.intersection(g):
                    groups.append(g)
    return groups


def ensure_arena(module):
    """Forward-entry hook of the model classes: the reference's scripts build the model, ``.to(device)`` it (or let
    ``wrap_model`` do so) and call it -- they know nothing of ``finalize``.  First call: place the parameters in a flat
    arena on the device they were moved to (bf16 compute copy when the call runs under ``torch.autocast``, the
    reference's ``--fp16`` switch: train_r2r.py:256-258, agent_base.py:195; fp32 otherwise).  Every call: honour a
    ``zero_grad(set_to_none=True)`` and a torch optimiser's parameter update (arena.py, torch-API bridge)."""
    arena = getattr(module, "arena", None)
    if arena is None:
        dev = next(module.parameters()).device
        if dev.type != "cuda":
            raise lib.BevBertHipError("the model has not been moved to the GPU: call .to('cuda') (or finalize(device, dtype)) "
                                      "first -- there is no CPU path")
        amp = torch.is_autocast_enabled()
        if amp:
            try:
                req = torch.get_autocast_dtype("cuda")
            except Exception:       # noqa: BLE001 -- older torch
                req = torch.get_autocast_gpu_dtype()
            if req != torch.bfloat16:
                # the reference's --fp16 is torch.cuda.amp.autocast() = float16 + GradScaler (train_r2r.py:226-227,256-258);
                # the MI355X path has one reduced-precision mode: bf16 compute copies over fp32 masters (no loss scaling
                # needed: GradScaler's scale / unscale_ / step still work and simply never find an overflow)
                import warnings
                warnings.warn(f"vln_bevbert_amd: autocast({req}) requested -- this build computes in bfloat16 with fp32 "
                              "master weights instead (same exponent range as fp32: no loss scaling required); a "
                              "GradScaler in the loop keeps working", RuntimeWarning, stacklevel=3)
        arena = finalize(module, dev, torch.bfloat16 if amp else torch.float32)
    elif arena.publish_grads:
        arena.maybe_lazy_zero()
        arena.maybe_refresh_shadow()
    # the residual-stream mode belongs to the model (finalize() of a second model in the process used to flip it
    # under every model finalized earlier): each forward entry installs its own arena's mode in the kernel library's
    # Python layer; the backward of that forward reads what its autograd nodes saved, not the switch
    ops.RT.res32 = bool(getattr(arena, "res32", False))
    return arena


def finalize(module, device, compute_dtype=torch.float32, residual=None):
    """Move a freshly built / loaded model into a ParamArena on ``device`` and cache the packed views.
    ``residual=torch.float32`` with bf16 compute: the post-norm blocks keep LayerNorm outputs and residual sums in fp32
    (torch.autocast's arithmetic, pretrain_src/train_r2r.py:256-258).  The mode is stored on the arena (``arena.res32``) and
    installed in the kernel library's Python layer (ops.RT.res32) at every forward entry (``ensure_arena``)."""
    if compute_dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be float32 or bfloat16")
    if residual not in (None, torch.float32, torch.bfloat16) or (residual == torch.bfloat16 and compute_dtype != torch.bfloat16):
        raise ValueError("residual stream dtype must be None (= compute dtype), float32, or bfloat16 with bf16 compute")
    ops.RT.res32 = compute_dtype == torch.bfloat16 and residual == torch.float32
    for b in module.buffers():
        b.data = b.data.to(device)
    arena = ParamArena(module, device, compute_dtype, groups=arena_groups(module))
    arena.res32 = ops.RT.res32
    for name, m in module.named_modules():
        if isinstance(m, _Finalizable):
            m._after_arena(arena, name + "." if name else "")
    module.arena = arena
    return arena
