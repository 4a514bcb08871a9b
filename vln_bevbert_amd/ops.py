"""Torch-facing wrappers over the C ABI (include/bevbert_hip.h) and their autograd Functions.

PyTorch here is plumbing: device memory, streams and the autograd tape.  The Linear layers stay on the vendor BLAS
(north star) but are issued to hipBLASLt directly through the C ABI (bevbert_gemm, cached per-shape plans).  Everything else on the hot path --
attention, bias/dropout/residual/LayerNorm, bias+GELU, the BEV splat, gmap aggregation, the optimiser -- is a
hand-written HIP kernel reached through ``lib.call``; none of them has a CPU or eager fallback.  The library GEMMs are
the one exception: a problem hipBLASLt's direct path cannot take (no algorithm for the layout, the plan budget is
spent, BEVBERT_LT_GEMM=0) is handed to torch's GEMM -- the same library underneath, 4x the host cost per call -- and
every such problem is reported once through ``warnings`` (``ops.GEMM_FALLBACKS`` counts them), never silently.

Gradient sinks: a parameter that lives in a ParamArena (arena.py) carries ``main_grad`` (fp32 view of the flat
gradient arena).  Backward kernels accumulate straight into that view and the Function returns ``None`` for the
parameter, so there are no per-parameter AccumulateGrad kernels, no bucket copies for the all-reduce, and the
optimiser sees one flat buffer.  Plain tensors (unit tests) get ordinary returned gradients.
"""
from . import lib  # noqa: F401
from .lib import call as _raw_call  # noqa: F401
from .lib import dtype_code, ptr, stream  # noqa: F401
from .ops_core import (  # noqa: F401
    ATTN_BITS, Branches, HEAD_DIM, RT, RowOfTable, _AttnBitsPlanner, _DROP_BITS_WORDS,
    _LT_WS_BYTES, _Runtime, _UseParam, _compute, _drop_bits_words, _f32, _gemm, _hash32,
    _mark_touched, _sink, call, use_param)
from .ops_reduce import (  # noqa: F401
    ReduceQueue, SCRATCH, ScratchRing, WgradStream, _PARTIAL_ROWS, _on_launch_stream, _partial_rows, join_captured_side_streams)
from .ops_gemm import (  # noqa: F401
    GEMM_FALLBACKS, GEMM_TUNING_FILE, HOIST_KV, _HoistedKV, _KVGradHolder, _LT_AUTOTUNE, _LT_ENABLED, _LT_PLANS,
    _LT_PLAN_BUDGET, _LT_RUN, _LT_UNSUPPORTED, _Linear, _LinearPacked, _PackedParam, _SPLITK_ENABLED, _SPLITK_MAX,
    _linear_dgrad, _linear_fwd, _linear_wgrad, _lt_gemm, _lt_ok, _param_grads, _rows, _split_k,
    _tuning_loaded, _warn_fallback, _wgrad_into, gemm_plan_count, hoisted_kv, linear, linear_packed, linear_packed_res,
    linear_res, load_gemm_tuning_table, save_gemm_tuning_table)
from .ops_rowops import (  # noqa: F401
    SegmentCSR, _BiasDropResLN, _BiasDropResLN32, _BiasGelu, _CrossEntropy, _DropoutAdd, _EmbedLN, _SapLoss,
    _SegmentWsum, bev_bin_points, bev_lift_bin, bev_splat_mean, bias_dropout_residual_layernorm, bias_dropout_residual_prenorm, bias_gelu, bias_relu, take_rows, weighted_mean, bce_rows, sem_select, bias_layernorm_plus, smallk_linear_layernorm_plus, smallk_linear_layernorm_plus_supported,
    cross_entropy_rows, dropout, dropout_keep_mask, embed_sum_layernorm, embedding_grad_small, layernorm, pixel_scale, sap_loss,
    sap_loss_supported, segment_wsum)
from .ops_attention import (  # noqa: F401
    _Attention, _strides, attention, attention_cross, attention_self, attn_drop_bits, graph_bias)
