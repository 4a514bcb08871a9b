"""ctypes binding of libbevbert_hip.so (C ABI: include/bevbert_hip.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).
There is deliberately NO fallback: if the library is missing or a call fails, an exception is raised --
this package never routes work to PyTorch eager kernels or to the CPU oracle.
"""
import ctypes
import os
import re
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbevbert_hip.so")
CSRC = os.path.join(_HERE, "csrc")
# Per-file compiler flags.  The persistent attention forward is compiled without the SLP vectoriser: it packs pairs of
# fp32 operations into v_pk_*_f32 instructions, and on gfx950 packed fp32 instructions do not overlap with the matrix
# instructions of the SIMD's other wave while plain ones do (scripts/probes/mfma_valu_overlap.hip,
# profiles/r05_mfma_valu_overlap_probe.txt).  Measured on that kernel: no difference either way.
EXTRA_FLAGS = {"attn_fwd4.hip": ["-fno-slp-vectorize"]}
SOURCES = ["splat.hip", "rowops.hip", "attn_simple.hip", "attn_f32.hip", "attn_mfma.hip", "attn_bwd1.hip", "attn_fwd2.hip", "attn_fwd4.hip", "attn_bwd2.hip", "attn_bwd3.hip", "attn_small.hip", "attn_short.hip", "smallk.hip", "sap_loss.hip", "graph_nav.hip", "gemm.hip", "capi.hip"]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "bevbert_hip.h")

F32, BF16, F16 = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}


class BevBertHipError(RuntimeError):
    pass


def dtype_code(t):
    try:
        return _DT[t.dtype if torch.is_tensor(t) else t]
    except KeyError:
        raise BevBertHipError(f"unsupported dtype {t}") from None


def build(force=False, verbose=False):
    """Compile csrc/*.hip into libbevbert_hip.so for gfx950 (cross-compiles without a GPU): one object per source
    (only the stale ones, in parallel), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    headers = [os.path.join(CSRC, h) for h in ("common.h", "attn_common.h", "attn_mfma_common.h", "attn_bwd7p1.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    objdir = os.path.join(_HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"]
    jobs, objs = [], []
    for name in SOURCES:
        src, obj = os.path.join(CSRC, name), os.path.join(objdir, name[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append([hipcc] + flags + EXTRA_FLAGS.get(name, []) + ["-c", src, "-o", obj])
    if not jobs and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-lhipblaslt"])
    return LIB_PATH


def header_symbols():
    with open(HEADER) as f:
        return sorted(set(re.findall(r"\b(bevbert_\w+)\s*\(", f.read())))


_lib = None
_P, _I, _I64, _U64, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float

_PROTOS = {
    "bevbert_bev_lift_bin": [_P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _F, _F, _P, _P, _P, _P],
    "bevbert_bev_bin_points": [_P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P],
    "bevbert_bev_splat_mean": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _I, _P],
    "bevbert_attn_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _F, _U64, _U64, _P, _I, _P],
    "bevbert_attn_drop_bits": [_P, _I, _I, _I, _I, _F, _U64, _U64, _P],
    "bevbert_attn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _F,
                         _U64, _U64, _P, _P],
    "bevbert_bias_dropout_residual_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _F, _U64,
                                                    _U64, _P],
    "bevbert_layernorm_post_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P],
    "bevbert_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U64, _I, _P],
    "bevbert_layernorm_res32_fwd": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _U64, _U64, _P],
    "bevbert_layernorm_res32_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _U64, _I, _I, _P],
    "bevbert_colsum_any": [_P, _P, _I, _I, _I, _I, _P],
    "bevbert_sem_select": [_P, _P, _I, _I, _I, _P, _P, _P, _P],
    "bevbert_bce_rows_fwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "bevbert_bce_rows_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "bevbert_weighted_mean_fwd": [_P, _P, _P, _F, _I, _P, _P],
    "bevbert_weighted_mean_bwd": [_P, _P, _P, _F, _I, _P, _P],
    "bevbert_layernorm_bwd_add": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U64, _I, _P],
    "bevbert_embed_sum_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _F, _U64, _U64, _P],
    "bevbert_embedding_grad": [_P, _P, _P, _I, _I, _I, _I, _P],
    "bevbert_rows_gather": [_P, _P, _P, _I, _I, _I, _P],
    "bevbert_graph_bias_fwd": [_P, _P, _P, _P, _I64, _P],
    "bevbert_graph_bias_bwd": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "bevbert_smallk_linear_layernorm_fwd": [_P] * 11 + [_I, _I, _I, _F, _I, _P],
    "bevbert_smallk_linear_layernorm_bwd": [_P] * 12 + [_I, _I, _I, _I, _P],
    "bevbert_rows_scatter": [_P, _P, _P, _I, _I, _I, _I, _P],
    "bevbert_embedding_grad_sliced": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "bevbert_bias_gelu_fwd": [_P, _P, _P, _I, _I, _I, _P],
    "bevbert_bias_gelu_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "bevbert_bias_relu_fwd": [_P, _P, _P, _I, _I, _I, _P],
    "bevbert_bias_relu_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "bevbert_colsum": [_P, _P, _P, _I, _I, _I, _I, _P],
    "bevbert_segment_wsum": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "bevbert_grad_norm_clip": [_P, _I64, _F, _F, _P, _P, _P],
    "bevbert_adamw_step": [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _F, _F, _F, _F, _F, _P],
    "bevbert_set_step_salt": [_P],
    "bevbert_colsum_partials": [_P, _P, _I, _I, _I, _P],
    "bevbert_multi_finalize": [_P, _I, _P],
    "bevbert_multi_accum": [_P, _I, _P],
    "bevbert_cast_f32": [_P, _P, _I64, _I, _P],
    "bevbert_zero": [_P, _I64, _P],
    "bevbert_accum_partials": [_P, _P, _I, _I64, _I, _P],
    "bevbert_dropout_keep_mask": [_P, _I64, _F, _U64, _U64, _P],
    "bevbert_gemm": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I64, _I64, _I64, _I, _I64, _I64, _I64, _I, _I, _I, _F, _P,
                     _I64, _I, _P],
    "bevbert_colsum_finalize": [_P, _I, _I, _I, _P, _P, _P, _I, _P],
    "bevbert_dropout_add": [_P, _P, _P, _I64, _I, _I, _F, _U64, _U64, _P],
    "bevbert_gm_update": [_P] * 10 + [_I, _I, _P, _P, _P],
    "bevbert_gm_nav_vars": [_P] * 7 + [_I, _I, _I] + [_P] * 7,
    "bevbert_gm_bev_select": [_P, _P, _I, _I, _P, _P, _P, _P, _P],
    "bevbert_gm_gather_views": [_P, _P, _P, _P, _I, _I64, _P],
    "bevbert_gm_embed_update": [_P] * 9 + [_I, _I, _I, _I, _P],
    "bevbert_gm_node_embeds": [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P],
    "bevbert_sap_loss_fwd": [_P] * 15 + [_I, _I, _I, _I, _I, _P],
    "bevbert_sap_loss_bwd": [_P] * 7 + [_I, _I, _I, _I, _P],
    "bevbert_cross_entropy_fwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "bevbert_cross_entropy_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "bevbert_gemm_run": [_I, _P, _P, _P, _P, _P, _I64, _P],
    "bevbert_gemm_run_add": [_I, _P, _P, _P, _P, _P, _P, _I64, _P],
}


def load():
    """dlopen the library and type every entry point. Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BevBertHipError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no PyTorch/CPU fallback for the hot path)")
    lib = ctypes.CDLL(LIB_PATH)
    lib.bevbert_last_error.restype = ctypes.c_char_p
    lib.bevbert_arch.restype = ctypes.c_char_p
    lib.bevbert_version.restype = _I
    lib.bevbert_attn_last_path.restype = ctypes.c_char_p
    lib.bevbert_attn_last_path.argtypes = [_I]
    lib.bevbert_hip_error_reset.restype = _I
    lib.bevbert_colsum_workspace_floats.restype = _I64
    lib.bevbert_colsum_workspace_floats.argtypes = [_I]
    lib.bevbert_gemm_plan_count.restype = _I
    lib.bevbert_gemm_rejected_count.restype = _I
    lib.bevbert_gemm_tuning_export.restype = _I64
    lib.bevbert_gemm_tuning_export.argtypes = [ctypes.c_char_p, _I64]
    lib.bevbert_gemm_tuning_import.restype = _I
    lib.bevbert_gemm_tuning_import.argtypes = [ctypes.c_char_p]
    lib.bevbert_attn_drop_bits_words.restype = _I64
    lib.bevbert_attn_drop_bits_words.argtypes = [_I, _I, _I, _I]
    lib.bevbert_smallk_workspace_floats.restype = _I64
    lib.bevbert_smallk_workspace_floats.argtypes = [_I, _I, _I]
    lib.bevbert_colsum_partial_rows.restype = _I
    lib.bevbert_colsum_partial_rows.argtypes = [_I]
    lib.bevbert_gemm_plan.restype = _I
    lib.bevbert_gemm_plan.argtypes = [_I, _I, _I, _I, _I, _I64, _I64, _I64, _I, _I64, _I64, _I64, _I, _I, _I, _I, _I64, _I]
    for name, args in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _I
    _lib = lib
    return lib


_fns = {}


def call(name, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise BevBertHipError(f"{name} failed ({rc}): {load().bevbert_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL). The tensor must live on a HIP device."""
    if t is None:
        return None
    if not t.is_cuda:
        raise BevBertHipError("libbevbert_hip.so operates on device memory only; got a CPU tensor "
                              "(no CPU fallback exists for the hot path)")
    return t.data_ptr()


_raw_stream = torch._C._cuda_getCurrentRawStream
_device_index = None


_override = None


def stream():
    """hipStream_t the next launch goes to: torch's current stream on this process's GPU, unless a launch-stream
    override is active (ops.WgradStream).  One process per GPU: the device index is read once;
    torch.cuda.current_stream() costs ~3 us of host time per call, the raw getter ~0.2 us."""
    global _device_index
    if _override is not None:
        return _override
    if _device_index is None:
        _device_index = torch.cuda.current_device()
    return _raw_stream(_device_index)


def set_stream_override(handle):
    """Route every following C-ABI launch to `handle` (a hipStream_t as int) until reset with None."""
    global _override
    _override = handle
