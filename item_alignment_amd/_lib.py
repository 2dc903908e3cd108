"""ctypes binding of libitemalign_hip.so (the C ABI declared in include/itemalign.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing, or an entry
point returns a non-zero code, this module raises.  Pointers are `tensor.data_ptr()` of live torch
tensors; the stream is torch's current HIP stream.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libitemalign_hip.so")

vp, i32, u32, f32, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_float, C.c_size_t


class LayerWeights(C.Structure):
    _fields_ = [(n, vp) for n in ("w_qkv", "b_qkv", "w_o", "b_o", "ln1_g", "ln1_b", "w_fc1", "b_fc1", "w_fc2", "b_fc2",
                                  "ln2_g", "ln2_b", "wt_qkv", "wt_o", "wt_fc1", "wt_fc2")]


class LayerGrads(C.Structure):
    _fields_ = [(n, vp) for n in ("w_qkv", "b_qkv", "w_o", "b_o", "ln1_g", "ln1_b", "w_fc1", "b_fc1", "w_fc2", "b_fc2",
                                  "ln2_g", "ln2_b")]


class LayerCfg(C.Structure):
    _fields_ = [("B", i32), ("L", i32), ("H", i32), ("I", i32), ("nh", i32), ("pre_ln", i32), ("eps", f32),
                ("hidden_drop", f32), ("attn_drop", f32), ("seed", u32), ("layer_id", u32), ("cu_seqlens", vp), ("total_tokens", i32),
                ("dx_colsum_out", vp), ("dy_colsum_done", i32), ("masked_rows_dead", i32)]


# name -> (restype, argtypes); must list every symbol include/itemalign.h declares
SIGNATURES = {
    "ia_strerror": (C.c_char_p, [i32]),
    "ia_abi_version": (i32, []),
    "ia_gemm_bf16": (i32, [vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, sz, vp]),
    "ia_gemm_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "ia_gemm_colsum_workspace_bytes": (sz, [i32, i32]),
    "ia_prof_begin": (i32, [i32, i32]),
    "ia_prof_end": (i32, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i32)]),
    "ia_prof_bytes": (C.c_double, []),
    "ia_debug_cu_hog": (i32, [i32, C.c_float, vp]),
    "ia_debug_gemm_dynamic": (i32, [i32]),
    "ia_ln_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, u32, u32, vp]),
    "ia_ln_bwd_workspace_bytes": (sz, [i32, i32]),
    "ia_ln_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, u32, u32, vp, sz, i32, vp]),
    "ia_ln_bwd2": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, u32, u32, vp, sz, i32, vp]),
    "ia_ln_bwd2_rows": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, u32, u32, vp, vp, sz, i32, vp]),
    "ia_colsum_workspace_bytes": (sz, [i32, i32]),
    "ia_colsum": (i32, [vp, i32, i32, i32, vp, i32, vp, sz, vp]),
    "ia_attn_fwd": (i32, [vp, vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, f32, f32, u32, vp]),
    "ia_attn_bwd": (i32, [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, u32, vp]),
    "ia_attn_bwd_bias_workspace_bytes": (sz, [i32, i32, i32]),
    "ia_attn_bwd_bias": (i32, [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, sz, i32, i32, i32, f32, f32, u32, vp]),
    "ia_attn_fwd_ps": (i32, [vp, vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, f32, f32, u32, vp]),
    "ia_attn_bwd_bias_ps": (i32, [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, sz, i32, i32, i32, f32, f32, u32, vp]),
    "ia_attn_bwd_bias_ex": (i32, [i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, sz, i32, i32, i32, f32, f32, u32, vp]),
    "ia_gemm_bf16_qscale": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i32, f32, vp]),
    "ia_attn_fwd_x": (i32, [vp, i32, vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, f32, f32, u32, vp]),
    "ia_attn_bwd_x": (i32, [vp, i32, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, f32, f32, u32, vp]),
    "ia_rotary_split_fwd": (i32, [vp, i32, vp, vp, i32, i32, i32, vp]),
    "ia_rotary_split_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "ia_swiglu_fwd": (i32, [vp, i32, vp, i32, i32, vp]),
    "ia_swiglu_bwd": (i32, [vp, vp, i32, vp, i32, i32, i32, vp]),
    "ia_kg_gather_fwd": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, i32, i32, i32, vp]),
    "ia_kg_gather_bwd": (i32, [vp, vp, i32, i32, vp, i32, i32, i32, vp]),
    "ia_kg_rows_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ia_kg_rows_bwd": (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, vp]),
    "ia_pair_sim_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "ia_pair_sim_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ia_nchw_to_nhwc_bf16": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ia_conv_nhwc_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32, i32, i32]),
    "ia_conv_nhwc_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "ia_conv_nhwc_bwd_data": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "ia_conv_nhwc_bwd_weight": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "ia_ws_conv_weight_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, vp]),
    "ia_ws_conv_weight_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "ia_silu_fwd": (i32, [vp, vp, sz, f32, vp]),
    "ia_silu_bwd": (i32, [vp, vp, vp, vp, sz, f32, vp]),
    "ia_silu_bwd_sum": (i32, [vp, vp, vp, vp, vp, sz, f32, vp]),
    "ia_avgpool2_fwd": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ia_avgpool2_bwd": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ia_gap_workspace_bytes": (sz, [i32, i32, i32]),
    "ia_gap_fwd": (i32, [vp, vp, i32, i32, i32, vp, sz, vp]),
    "ia_gap_bwd": (i32, [vp, vp, i32, i32, i32, vp]),
    "ia_eca_fwd": (i32, [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, vp, sz, vp]),
    "ia_eca_fwd_linear_workspace_bytes": (sz, [i32, i32, i32]),
    "ia_eca_fwd_linear": (i32, [vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, f32, vp, vp, i32, i32, i32, f32, vp, sz, vp]),
    "ia_eca_silu_bwd": (i32, [vp, vp, vp, vp, f32, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp, sz, vp]),
    "ia_eca_bwd_workspace_bytes": (sz, [i32, i32, i32]),
    "ia_eca_bwd": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, vp, sz, vp]),
    "ia_conv3x3_padded_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ia_conv3x3_padded_bwd_data": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ia_conv3x3_direct_supported": (i32, [i32, i32, i32]),
    "ia_conv3x3_flip_weights": (i32, [vp, vp, i32, i32, i32, vp]),
    "ia_conv3x3_padded_bwd_data_t": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ia_conv3x3_padded_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "ia_conv3x3_padded_bwd_weight": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "ia_pad_rows": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ia_silu_pad_fwd": (i32, [vp, vp, i32, i32, i32, i32, f32, i32, i32, vp]),
    "ia_silu_pad_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, vp]),
    "ia_bn_act_workspace_bytes": (sz, [i32, i32, i32]),
    "ia_bn_act_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, i32, i32, vp, sz, vp]),
    "ia_bn_act_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]),
    "ia_patches_nchw": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ia_conv3x3_s2_supported": (i32, [i32, i32, i32]),
    "ia_conv3x3_s2_padded_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "ia_conv3x3_s2_padded_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ia_conv3x3_s2_padded_bwd_data": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "ia_conv3x3_s2_padded_bwd_weight": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "ia_conv3x3_s2_dgrad_supported": (i32, [i32, i32, i32]),
    "ia_conv3x3_s2_padded_bwd_data_t": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ia_gn_act_workspace_bytes": (sz, [i32, i32, i32]),
    "ia_gn_act_fwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, sz, vp]),
    "ia_gn_act_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]),
    "ia_maxpool3s2_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "ia_maxpool3s2_fwd_ex": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ia_maxpool3s2_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "ia_rows_subsample_fwd": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ia_rows_subsample_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ia_conv_weight_pack": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ia_conv_weight_unpack_grad": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ia_resize_pass_u8": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ia_resize_pass_u8_ex": (i32, [vp, sz, sz, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ia_color_jitter_step_u8": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ia_u8_to_nchw_normalized": (i32, [vp, vp, vp, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp]),
    "ia_embed_ln_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, u32, u32, vp]),
    "ia_embed_ln_bwd_workspace_bytes": (sz, [i32, i32]),
    "ia_embed_ln_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, u32, u32,
                               vp, sz, vp]),
    "ia_im2col_patch": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ia_vit_tokens_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "ia_vit_tokens_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ia_gather_rows_fwd": (i32, [vp, i32, vp, vp, i32, i32, f32, u32, u32, vp]),
    "ia_gather_rows_bwd": (i32, [vp, i32, vp, vp, i32, i32, f32, u32, u32, i32, vp]),
    "ia_linear_small_fwd": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ia_linear_small_bwd": (i32, [vp, vp, vp, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp]),
    "ia_pair_head_ce_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ia_attn_fwd_varlen": (i32, [vp, vp, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, f32, f32, u32, vp]),
    "ia_attn_bwd_varlen": (i32, [vp, vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, u32, vp]),
    "ia_attn_fwd_varlen_ps": (i32, [vp, vp, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, f32, f32, u32, vp]),
    "ia_attn_bwd_varlen_ps": (i32, [vp, vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, u32, vp]),
    "ia_span_mean_fwd": (i32, [vp, i32, vp, vp, i32, i32, vp]),
    "ia_span_mean_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "ia_pair_head_ce_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ia_transpose_bf16_batched": (i32, [vp, vp, vp, i32, i32, vp]),
    "ia_adamw_flat": (i32, [vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, f32, i32, f32, vp]),
    "ia_cast_f32_to_bf16": (i32, [vp, vp, sz, vp]),
    "ia_cast_bf16_to_f32": (i32, [vp, vp, sz, vp]),
    "ia_layer_stash_bytes": (sz, [C.POINTER(LayerCfg)]),
    "ia_layer_bwd_scratch_bytes": (sz, [C.POINTER(LayerCfg)]),
    "ia_layer_fwd": (i32, [C.POINTER(LayerCfg), C.POINTER(LayerWeights), vp, vp, vp, vp, vp]),
    "ia_comm_unique_id": (i32, [vp]),
    "ia_comm_init": (i32, [vp, i32, i32, C.POINTER(vp)]),
    "ia_comm_allreduce_bucket": (i32, [vp, vp, sz, i32, vp]),
    "ia_comm_finalize": (i32, [vp]),
    "ia_comm_last_error": (C.c_char_p, []),
    "ia_layer_infer_scratch_bytes": (sz, [C.POINTER(LayerCfg)]),
    "ia_layer_fwd_infer": (i32, [C.POINTER(LayerCfg), C.POINTER(LayerWeights), vp, vp, vp, vp, sz, vp]),
    "ia_layer_bwd": (i32, [C.POINTER(LayerCfg), C.POINTER(LayerWeights), C.POINTER(LayerGrads), vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "ia_layer_bwd2": (i32, [C.POINTER(LayerCfg), C.POINTER(LayerWeights), C.POINTER(LayerGrads), vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
}

_lib = None


ABI_VERSION = 10      # = IA_ABI_VERSION of include/itemalign.h (tests/test_cabi_symbols.py keeps the two in step)


class ItemAlignError(RuntimeError):
    pass


def load():
    """Load the HIP extension; raise if it is absent (there is deliberately no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ItemAlignError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C item_alignment_amd/csrc`). The MI355X path has no CPU fallback.")
    # torch must be imported first: it brings its own libamdhip64, and the extension has to bind to THAT runtime
    # instance (loading /opt/rocm's copy first leaves the process with a runtime that sees no device)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    got = lib.ia_abi_version()
    if got != ABI_VERSION:
        raise ItemAlignError(f"{LIB_PATH} has ABI version {got}, this package needs {ABI_VERSION} (include/itemalign.h: IA_ABI_VERSION): "
                             "stale build, run `make -C item_alignment_amd/csrc`")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ia_strerror(rc).decode()
        raise ItemAlignError(f"{what} failed: {msg} (code {rc})")


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """data_ptr of a tensor or None -> NULL."""
    return None if t is None else t.data_ptr()
