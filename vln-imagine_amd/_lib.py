"""ctypes binding of libvlni.so (the C-ABI declared in include/vlni.h).

The product path has no CPU fallback: if the library is missing or a call fails,
this module raises. Pointers are passed as integers (tensor.data_ptr()).
"""
import ctypes
import os

import torch  # noqa: F401  MUST precede the dlopen below: libvlni.so needs libamdhip64.so.7 by SONAME and has to
#                            bind to the copy torch already loaded (two HIP runtimes in one process cannot both
#                            own the device: the second reports "no ROCm-capable device").

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLNI_LIB_PATH") or os.path.join(_HERE, "libvlni.so")   # override: A/B runs against another build

P, L, I, F, U = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_uint

# name -> argtypes, in the order of include/vlni.h
SIGNATURES = {
    "vlni_gemm_nt": [I, P, L, P, L, P, L, I, I, I, P, I, P, L, P, L, P, L, I, F, I, I, P],
    "vlni_gemm_nt_v": [I, P, L, P, L, P, L, I, I, I, P, I, P, L, P, L, P, L, I, F, I, I, I, F, U, P],
    "vlni_gemm_nt_dual": [I, P, P, P, P, P, P, P, I, I, P, I, P, P, P, P, P, P, I, I, F, P, P],
    "vlni_gemm_nt_multi": [I, I, P, P, P, P, P, P, P, I, I, P, I, P, P, P, P, P, P, I, I, F, P, P],
    "vlni_gemm_tn_bf16": [P, L, P, L, P, L, I, I, I, P, I, P],
    "vlni_gemm_tn_bf16_grouped": [I, P, P, P, L, L, P, L, I, I, P, I, P],
    "vlni_gemm_tn_bf16_grouped_v": [I, P, P, P, L, L, P, L, I, I, P, I, I, P],
    "vlni_gemm_tn_bf16_grouped_part": [I, P, P, P, L, L, P, L, I, I, P, I, I, P],
    "vlni_reduce_parts": [P, I, I, P],
    "vlni_reduce_parts_sq": [P, I, I, P, I, P],
    "vlni_sumsq_fold": [P, I, P, P],
    "vlni_upload": [P, P, L, P],
    "vlni_shadow_refresh": [I, P, I, I, P],
    "vlni_gemm_tn_h16_grouped_v": [I, I, P, P, P, L, L, P, L, I, I, P, I, I, P],
    "vlni_gemm_tn_h16_grouped_part": [I, I, P, P, P, L, L, P, L, I, I, P, I, I, P],
    "vlni_attn_fwd": [I, P, L, P, L, P, L, P, P, P, L, P, I, I, I, I, F, F, U, P],
    "vlni_attn_probs": [I, P, L, P, L, P, P, P, I, I, I, I, F, P],
    "vlni_attn_bwd": [I, P, L, P, L, P, L, P, P, P, L, P, L, P, P, L, P, L, P, L, P, I, I, I, I, F, F, U, P],
    "vlni_attn_fwd_dual": [I, P, P, P, P, P, P, P, P, P, P, P, I, I, P, P, F, F, P, P],
    "vlni_attn_bwd_dual": [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, P, P, F, F, P, P],
    "vlni_layernorm_fwd": [I, P, L, P, P, F, P, L, P, P, I, I, P],
    "vlni_layernorm_bwd": [I, P, L, P, L, P, P, P, P, L, P, P, I, I, P, L, P, L, F, U, P],
    "vlni_layernorm_fwd_dual": [I, P, P, P, P, F, P, P, P, P, P, I, P],
    "vlni_layernorm_bwd_dual": [I, P, P, P, P, P, P, P, P, P, P, P, P, I, P, P, P, P, F, P, P],
    "vlni_sum_layernorm_fwd": [I, I, P, P, P, P, P, P, F, P, L, P, L, P, P, I, I, P],
    "vlni_bias_residual_layernorm_fwd": [I, P, L, P, P, L, P, P, F, P, L, P, L, P, P, I, I, P],
    "vlni_bias_residual_layernorm_bwd": [I, P, L, P, L, P, P, P, P, L, P, P, P, I, I, P],
    "vlni_cast": [I, I, P, P, L, P],
    "vlni_transpose": [I, I, P, L, P, L, I, I, I, P],
    "vlni_colsum": [I, P, L, I, I, P, P],
    "vlni_smallk_linear_fwd": [I, P, L, P, P, P, L, I, I, I, P],
    "vlni_smallk_linear_bwd": [I, P, L, P, L, P, P, I, I, I, P],
    "vlni_scatter_add_rows": [I, P, L, P, P, I, I, I, P],
    "vlni_scatter_add_rows_small": [I, P, L, P, P, I, I, I, P],
    "vlni_seqmean_fwd": [I, P, P, P, I, I, I, P],
    "vlni_seqmean_bwd": [I, P, P, P, I, I, I, P],
    "vlni_gate_rows_fwd": [I, P, P, P, I, I, I, I, I, I, P],
    "vlni_gate_rows_bwd": [I, P, P, P, P, P, I, I, I, I, I, I, P],
    "vlni_rowdot_fwd": [I, P, L, P, P, P, P, I, I, P],
    "vlni_rowdot_bwd": [I, P, P, L, P, P, P, L, P, P, I, I, P],
    "vlni_cross_entropy": [P, L, P, L, P, P, L, I, I, P],
    "vlni_segment_mean_fwd": [I, P, L, P, P, P, I, I, P],
    "vlni_segment_mean_bwd": [I, P, P, P, P, I, I, P],
    "vlni_cosine_fwd": [I, P, P, F, P, P, P, I, I, P],
    "vlni_cosine_bwd": [I, P, P, P, P, P, P, P, P, I, I, P],
    "vlni_pairdot_fwd": [P, P, P, I, I, I, P],
    "vlni_pairdot_bwd": [P, P, P, P, P, I, I, I, P],
    "vlni_dropout": [I, P, P, L, F, U, P],
    "vlni_act_bwd": [I, I, P, P, P, L, P],
    "vlni_adamw_step": [P, P, P, P, P, L, F, F, F, F, F, I, P, P],
    "vlni_sumsq": [P, L, P, P],
    "vlni_clip_coef": [P, F, P, P],
    "vlni_transpose_batched": [I, P, I, I, P],
    "vlni_set_dropout_seed_base": [P],
    "vlni_set_index_error_counter": [P],
    "vlni_build_views": [I, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    "vlni_graph_init": [P, P, P, I, I, P],
    "vlni_graph_observe": [P, P, P, P, P, P, P, P, P, P, I, I, I, P],
    "vlni_graph_pos_fts": [P, P, P, P, P, P, P, P, L, L, P, I, I, I, I, P],
    "vlni_graph_pair_dists": [P, P, P, I, I, I, P],
    "vlni_gather_rows_or_zero": [I, P, L, P, P, I, I, P],
    "vlni_duet_fuse_fwd": [P, P, P, P, P, I, I, I, P],
    "vlni_duet_fuse_bwd": [P, P, P, P, I, I, I, P],
    "vlni_duet_heads_fwd": [P] * 11 + [I, I, I, P],
    "vlni_duet_heads_bwd": [P] * 14 + [I, I, I, P],
    "vlni_optim_prepare": [P, F, F, F, P, P],
    "vlni_adamw_step_dev": [P, P, P, P, P, L, P, F, F, F, F, P, P],
    "vlni_optim_prepare_groups": [P, F, F, F, P, P, P, I, I, P],
    "vlni_adamw_step_groups": [P, P, P, P, P, I, L, P, P, P, I, F, F, F, F, P, P],
    "vlni_scale_cast": [I, I, P, P, L, F, P],
    "vlni_embed_combine_fwd": [I, P, L, P, P, P, L, I, P, P, P, P, P, L, P, P, P, I, P, P, I, P, P, F, P, P, P, L, P, P, P, P, P, P, F, U, I, I, P],
    "vlni_ln_rowdot_fwd": [I, P, L, P, P, F, P, P, P, P, P, P, P, F, U, I, I, P],
    "vlni_self_att_block_fwd": [P, P], "vlni_self_att_block_bwd": [P, P], "vlni_ffn_block_fwd": [P, P], "vlni_ffn_block_bwd": [P, P],
}


class VlniError(RuntimeError):
    pass


_lib = None


def load():
    """Loads libvlni.so (built by vln_imagine_amd.build / __graft_entry__.build). Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VlniError(
            f"{LIB_PATH} not found: the HIP operator library is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback).")
    lib = ctypes.CDLL(LIB_PATH)
    lib.vlni_last_error.restype = ctypes.c_char_p
    lib.vlni_last_error.argtypes = []
    lib.vlni_version.restype = ctypes.c_int
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    _lib = lib
    return lib


_fn = {}


def call(name, *args):
    f = _fn.get(name)
    if f is None:
        f = _fn[name] = getattr(load(), name)
    rc = f(*args)
    if rc != 0:
        raise VlniError(f"{name} failed ({rc}): {load().vlni_last_error().decode()}")
