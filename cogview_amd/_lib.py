"""ctypes binding of libcogview_hip.so (see include/cogview_hip.h).

There is NO CPU fallback: if the shared library cannot be loaded every compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COGVIEW_HIP_LIB points the loader at another build of the same ABI (kernel experiments); default: in-tree
LIB_PATH = os.environ.get("COGVIEW_HIP_LIB") or os.path.join(_HERE, "lib", "libcogview_hip.so")

F16, BF16, F32 = 0, 1, 2
EPI_BIAS, EPI_GELU, EPI_DGELU, EPI_DROPOUT, EPI_ABSMAX, EPI_ACCUM, EPI_COLSUM = 1, 2, 4, 8, 16, 32, 64
EPI_GELU_DAUX, EPI_MULAUX = 128, 256

_ERR = {1: "bad argument (shape / alignment / dtype)", 2: "kernel launch failure", 3: "unsupported combination"}


class CogviewHipError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("trans_a", C.c_int), ("trans_b", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("A", C.c_void_p), ("lda", C.c_int),
        ("B", C.c_void_p), ("ldb", C.c_int),
        ("C", C.c_void_p), ("ldc", C.c_int),
        ("out_f32", C.c_int), ("flags", C.c_int),
        ("bias", C.c_void_p),
        ("aux", C.c_void_p), ("ldaux", C.c_int),
        ("absmax", C.c_void_p),
        ("dropout_p", C.c_float), ("seed", C.c_uint64), ("stream_id", C.c_uint64),
        ("splitk", C.c_int), ("kernel_variant", C.c_int),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("colsum_partial", C.c_void_p),
        ("dropout_row0", C.c_longlong),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("B", C.c_int), ("H", C.c_int), ("s_q", C.c_int), ("s_k", C.c_int),
        ("head_dim", C.c_int), ("sep", C.c_int),
        ("scale", C.c_float), ("dropout_p", C.c_float), ("seed", C.c_uint64), ("stream_id", C.c_uint64),
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
        ("dout", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
        ("lse", C.c_void_p), ("dvec", C.c_void_p),
        ("q_bs", C.c_longlong), ("k_bs", C.c_longlong), ("v_bs", C.c_longlong), ("o_bs", C.c_longlong),
        ("do_bs", C.c_longlong), ("dq_bs", C.c_longlong), ("dk_bs", C.c_longlong), ("dv_bs", C.c_longlong),
        ("q_rs", C.c_int), ("k_rs", C.c_int), ("v_rs", C.c_int), ("o_rs", C.c_int),
        ("do_rs", C.c_int), ("dq_rs", C.c_int), ("dk_rs", C.c_int), ("dv_rs", C.c_int),
        ("colsum_partial", C.c_void_p),
        ("kv_index", C.c_void_p), ("kv_index_bs", C.c_longlong),
        ("kv_index_gs", C.c_longlong), ("sparse_window", C.c_int), ("sparse_pivots", C.c_int), ("sparse_pivot_bias", C.c_float),
        ("keep_bits", C.c_void_p),
        ("mask", C.c_void_p), ("mask_bs", C.c_longlong),
    ]


class LnPrologue(C.Structure):
    _fields_ = [
        ("z", C.c_void_p), ("z_absmax", C.c_void_p),
        ("gamma_post", C.c_void_p), ("beta_post", C.c_void_p), ("residual", C.c_void_p), ("t_out", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("eps", C.c_float), ("stream_f32", C.c_int),
    ]


class AttnDecodeDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("B", C.c_int), ("H", C.c_int), ("capacity", C.c_int), ("head_dim", C.c_int),
        ("scale", C.c_float),
        ("qkv", C.c_void_p), ("qkv_bs", C.c_longlong),
        ("cache", C.c_void_p), ("cache_bs", C.c_longlong), ("cache_rs", C.c_int),
        ("out", C.c_void_p), ("out_bs", C.c_longlong),
        ("pos", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("skip_combine", C.c_int),
    ]


class AdamDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int),
        ("params", C.c_void_p), ("grads", C.c_void_p),
        ("master", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
        ("chunk_start", C.c_void_p), ("chunk_len", C.c_void_p), ("chunk_group", C.c_void_p), ("nchunks", C.c_int),
        ("lr", C.c_float * 8), ("weight_decay", C.c_float * 8),
        ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("step", C.c_int), ("bias_correction", C.c_int), ("adam_w_mode", C.c_int),
        ("beta1_d", C.c_double), ("beta2_d", C.c_double),
        ("inv_loss_scale", C.c_float), ("max_grad_norm", C.c_float),
        ("stats", C.c_void_p), ("norm_sumsq_override", C.c_void_p),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("B", C.c_int), ("IH", C.c_int), ("IW", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
        ("relu", C.c_int),
        ("in", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
        ("rgb_w", C.c_void_p), ("rgb_partial", C.c_void_p),
        ("relu_in", C.c_int), ("residual", C.c_void_p), ("relu_residual", C.c_int),
    ]


CONV_4X4_S2, CONV_1X1, CONVT_4X4_S2, CONV_3X3_S1 = 0, 1, 2, 3

_vp, _i, _f, _u64, _i64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_int64, C.c_size_t

# name -> (restype, argtypes).  Mirrors include/cogview_hip.h one-to-one (tests/test_abi.py checks this).
SIGNATURES = {
    "cogv_version": (_i, []),
    "cogv_arch": (C.c_char_p, []),
    "cogv_gemm": (_i, [C.POINTER(GemmDesc), _vp]),
    "cogv_gemm_workspace_bytes": (_sz, [C.POINTER(GemmDesc)]),
    "cogv_gemm_pick_splitk": (_i, [_i, _i, _i]),
    "cogv_gemm_grouped": (_i, [C.POINTER(GemmDesc), _i, _vp]),
    "cogv_gemm_colsum_rows": (_i, [_i]),
    "cogv_colsum_finalize": (_i, [_i, _vp, _i, _i, _vp, _i, _vp]),
    "cogv_gemm_pick_splitk_tiles": (_i, [_i, _i]),
    "cogv_sandwich_ln_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp]),
    "cogv_sandwich_ln_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _u64, _u64,
                                  _vp, _sz, _i, _vp]),
    "cogv_sandwich_ln_bwd_marked": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _sz, _i, _vp]),
    "cogv_ln_bwd_workspace_bytes": (_sz, [_i, _i]),
    "cogv_sandwich_ln_bwd_pair": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f,
                                       _vp, _sz, _vp]),
    "cogv_ln_bwd_pair_workspace_bytes": (_sz, [_i, _i]),
    "cogv_ln_bwd_num_blocks": (_i, [_i]),
    "cogv_gemm_reserve_cus": (_i, [_i]),
    "cogv_attention_fwd": (_i, [C.POINTER(AttnDesc), _vp]),
    "cogv_attention_bwd": (_i, [C.POINTER(AttnDesc), _vp]),
    "cogv_gemv_ln": (_i, [C.POINTER(GemmDesc), C.POINTER(LnPrologue), _vp]),
    "cogv_gemv_attn": (_i, [C.POINTER(GemmDesc), _vp, _i, _i, _vp]),
    "cogv_attention_decode": (_i, [C.POINTER(AttnDecodeDesc), _vp]),
    "cogv_attention_decode_workspace_bytes": (_sz, [_i, _i, _i]),
    "cogv_attention_keep_bits_bytes": (_sz, [_i, _i, _i, _i]),
    "cogv_sparse_slot_reduce": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cogv_embedding_fwd": (_i, [_i, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i, _f, _u64, _u64, _i, _vp]),
    "cogv_embedding_bwd": (_i, [_i, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _i64, _i, _f, _u64, _u64, _vp, _sz, _i, _vp]),
    "cogv_embedding_bwd_workspace_bytes": (_sz, [_i64, _i64]),
    "cogv_gelu_fwd": (_i, [_i, _vp, _vp, _sz, _vp]),
    "cogv_gelu_bwd": (_i, [_i, _vp, _vp, _vp, _sz, _vp]),
    "cogv_dropout": (_i, [_i, _vp, _vp, _sz, _f, _u64, _u64, _vp, _vp]),
    "cogv_add": (_i, [_i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "cogv_add_stream": (_i, [_i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "cogv_scale": (_i, [_i, _vp, _vp, _sz, _f, _vp]),
    "cogv_absmax": (_i, [_i, _vp, _sz, _vp, _vp]),
    "cogv_colsum": (_i, [_i, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "cogv_colsum_workspace_bytes": (_sz, [_i, _i]),
    "cogv_ce_fwd": (_i, [_i, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "cogv_ce_bwd": (_i, [_i, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "cogv_grad_stats": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "cogv_grad_stats_workspace_bytes": (_sz, []),
    "cogv_adamw_step": (_i, [C.POINTER(AdamDesc), _vp]),
    "cogv_cast_flat": (_i, [_i, _vp, _vp, _sz, _vp]),
    "cogv_cast_flat_back": (_i, [_i, _vp, _vp, _sz, _vp]),
    "cogv_conv2d_nhwc_f32": (_i, [C.POINTER(ConvDesc), _vp]),
    "cogv_rgb_finalize_f32": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
    "cogv_vq_argmin_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "cogv_nchw3_to_nhwc4_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "cogv_embed_code_f32": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "cogv_conv1x1_to_rgb_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
}

_lib = None


def lib():
    """Return the loaded library, loading it on first use.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CogviewHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  cogview_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError here == ABI drift, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise CogviewHipError(f"{what} failed: {_ERR.get(rc, rc)}")
