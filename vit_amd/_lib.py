"""ctypes binding of libvtcnn_hip.so (C ABI: include/vtcnn_hip.h).  Fails loudly if missing."""
import ctypes as C
import os

# PyTorch-ROCm wheels bundle their own HIP runtime.  It has to be the first (and only) HIP runtime
# mapped into the process: loading libvtcnn_hip.so first would pull in the system libamdhip64 and
# the two runtimes would then fight over the device ("no ROCm-capable device is detected").
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvtcnn_hip.so")

VT_NAME_LEN = 40
VT_MAX_TENSORS = 96


class VtDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("image_channels", "input_dim", "tokens", "token_channels",
                                       "vt_channels", "heads", "num_classes")]


class VtTensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * VT_NAME_LEN), ("offset", C.c_int), ("size", C.c_int), ("ndim", C.c_int),
                ("shape", C.c_int * 4), ("kind", C.c_int)]


class VtTrainState(C.Structure):
    _fields_ = [("version", C.c_int), ("epoch", C.c_int), ("step", C.c_longlong), ("loader_seed", C.c_uint),
                ("loader_resets", C.c_int)]


class VtGemmTerm(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("strideA", C.c_long), ("strideB", C.c_long), ("K", C.c_int),
                ("lda", C.c_int), ("ldb", C.c_int), ("transB", C.c_int)]


class VtLayout(C.Structure):
    _fields_ = [("n_tensors", C.c_int), ("n_trainable", C.c_int), ("n_total", C.c_int),
                ("n_trainable_exact", C.c_int), ("t", VtTensorInfo * VT_MAX_TENSORS)]


# name -> (restype, argtypes); every symbol declared in include/vtcnn_hip.h
p, i, l, f, u = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint
SIGNATURES = {
    "vt_error_string": (C.c_char_p, [i]),
    "vt_version": (i, []),
    "vt_model_create": (i, [C.POINTER(VtDims), i, i, p, p, C.POINTER(p)]),
    "vt_model_destroy": (i, [p]),
    "vt_model_layout": (i, [p, C.POINTER(VtLayout)]),
    "vt_model_is_fused": (i, [p]),
    "vt_model_set_fused_reductions": (i, [p, i]),
    "vt_model_train_step": (i, [p, p, i, p, C.c_float, C.c_float, p, p]),
    "vt_model_train_step_grouped": (i, [p, p, i, i, p, C.c_float, C.c_float, p, p]),
    "vt_model_init_seed": (i, [p, u]),
    "vt_model_set_arena": (i, [p, p]),
    "vt_model_get_arena": (i, [p, p]),
    "vt_model_get_grads": (i, [p, p]),
    "vt_model_arena_ptr": (p, [p]),
    "vt_model_grads_ptr": (p, [p]),
    "vt_model_save": (i, [p, C.c_char_p]),
    "vt_model_load": (i, [p, C.c_char_p]),
    "vt_model_set_matmul_precision": (i, [p, i]),
    "vt_model_set_training": (i, [p, i]),
    "vt_model_set_concurrency": (i, [p, i]),
    "vt_model_set_deferred_attention": (i, [p, i]),
    "vt_model_attention_join": (i, [p, p]),
    "vt_model_keep_internals": (i, [p, i]),
    "vt_model_forward": (i, [p, p, i, p, p]),
    "vt_model_forward_grouped": (i, [p, p, i, i, p, p]),
    "vt_model_backward": (i, [p, p, p]),
    "vt_model_backward_main": (i, [p, p, p]),
    "vt_model_backward_attention": (i, [p, p, p]),
    "vt_model_attention_range": (i, [p, p, p]),
    "vt_model_sgd_step": (i, [p, f, f, p]),
    "vt_model_adam_step": (i, [p, f, f, f, f, f, p]),
    "vt_model_sgd_step_range": (i, [p, f, f, i, i, p]),
    "vt_model_metrics": (i, [p, p, p, p, p]),
    "vt_model_graph_capture": (i, [p, p, p, i, f, f]),
    "vt_model_graph_capture_forward": (i, [p, p, i, p]),
    "vt_model_graph_launch": (i, [p, p]),
    "vt_model_read": (l, [p, C.c_char_p, p, l]),
    "vt_model_profile": (i, [p, C.c_char_p]),
    "vt_model_profile_read": (i, [p, i, C.c_char_p, i, C.POINTER(f), C.POINTER(i)]),
    "vt_model_profile_timeline": (i, [p, i, C.c_char_p, i, C.POINTER(f), C.POINTER(f)]),
    "vt_host_init_arena": (i, [C.POINTER(VtDims), u, p]),
    "vt_checkpoint_write": (i, [C.POINTER(VtDims), p, C.c_char_p]),
    "vt_checkpoint_read": (i, [C.POINTER(VtDims), p, C.c_char_p]),
    "vt_checkpoint_write_v2": (i, [C.POINTER(VtDims), p, C.POINTER(VtTrainState), C.c_char_p]),
    "vt_checkpoint_read_v2": (i, [C.POINTER(VtDims), p, C.POINTER(VtTrainState), C.c_char_p]),
    "vt_model_save_v2": (i, [p, C.POINTER(VtTrainState), C.c_char_p]),
    "vt_model_load_v2": (i, [p, C.POINTER(VtTrainState), C.c_char_p]),
    "vt_idx_read_images": (i, [C.c_char_p, i, p, C.POINTER(i), C.POINTER(i), C.POINTER(i)]),
    "vt_image_read_gray": (i, [C.c_char_p, p, i, p, p]),
    "vt_idx_read_labels": (i, [C.c_char_p, i, p, C.POINTER(i)]),
    "vt_loader_indices": (i, [i, u, i, p]),
    "vt_conv3x3_fwd": (i, [p, p, p, p, p, i, i, i, i, p]),
    "vt_bn_stats_finalize": (i, [p, i, i, l, l, p, p, p, p]),
    "vt_bn_stats_finalize_grouped": (i, [p, i, i, i, l, l, p, p, p, p, p]),
    "vt_bn_stats_eval": (i, [p, p, i, p, p]),
    "vt_transformer_ffn_fwd_grouped": (i, [p] * 12 + [l, l, p]),
    "vt_projector_tokens_fwd_grouped": (i, [p] * 10 + [i, i, p]),
    "vt_head_fwd_grouped": (i, [p] * 10 + [i, i, i, i, p]),
    "vt_tokenizer_fwd": (i, [p] * 12 + [i, i, p]),
    "vt_transformer_ffn_fwd": (i, [p] * 12 + [l, p]),
    "vt_projector_tokens_fwd": (i, [p] * 10 + [i, p]),
    "vt_projector_fwd": (i, [p] * 11 + [i, i, p]),
    "vt_head_fwd": (i, [p] * 10 + [i, i, i, p]),
    "vt_metrics": (i, [p, p, i, i, p, p, p]),
    "vt_loader_gather": (i, [p, p, p, i, i, p, p, p]),
    "vt_eval_accumulate": (i, [p, p, i, i, i, p, p, p, p, p]),
    "vt_eval_accumulate_grouped": (i, [p, p, i, i, i, i, p, p, p, p, p]),
    "vt_head_bwd": (i, [p] * 11 + [i, i, i, p]),
    "vt_fc_wgrad": (i, [p] * 5 + [i, i, i, p]),
    "vt_fc_wgrad_static": (i, [p] * 4 + [i, i, i, p]),
    "vt_dp_init": (i, [i, p, p]),
    "vt_dp_destroy": (i, [p]),
    "vt_dp_ranks": (i, [p]),
    "vt_dp_allreduce_async": (i, [p, p, l, l, p]),
    "vt_dp_allreduce_rank_async": (i, [p, i, p, l, l, p]),
    "vt_dp_exchange_begin": (i, [p, i, p, l, l, p]),
    "vt_dp_exchange_join": (i, [p, i, p]),
    "vt_dp_wait": (i, [p]),
    "vt_bn_bwd_finalize": (i, [p, i, i, l, p, p, p, p, p, p, p]),
    "vt_bn_bwd_finalize_grouped": (i, [p, i, i, i, l, l, p, p, p, p, p, p]),
    "vt_projector_bwd": (i, [p] * 13 + [i, i, p]),
    "vt_projector_bwd_windows": (i, [p] * 13 + [i, i, p]),
    "vt_head_bwd_windows_blocks": (i, [i]),
    "vt_head_bwd_windows": (i, [p] * 12 + [i, i, i, p]),
    "vt_head_expand_windows": (i, [p, p, i, i, p]),
    "vt_dense_bwd_rows": (i, [p, p, p, p, p, p, i, p, l, i, i, i, p]),
    "vt_tokenizer_bwd": (i, [p] * 11 + [i, i, p]),
    "vt_conv3x3_wgrad": (i, [p, p, p, p, i, i, i, i, p]),
    "vt_reduce_rows": (i, [p, l, i, i, p, p]),
    "vt_gemm_f32": (i, [i, i, i, i, i, f, p, i, C.c_long, p, i, C.c_long, i, p, i, C.c_long, p, i, p, C.c_long, p]),
    "vt_gemm_bf16_mfma": (i, [i, i, i, i, i, f, p, i, C.c_long, p, i, C.c_long, i, p, i, C.c_long, p, i, p, C.c_long, p]),
    "vt_gemm_f32x3": (i, [i, i, i, i, i, f, p, i, C.c_long, p, i, C.c_long, i, p, i, C.c_long, p, i, p, C.c_long, p]),
    "vt_gemm_sum": (i, [i, i, C.POINTER(VtGemmTerm), i, i, f, i, p, i, C.c_long, p, i, p]),
    "vt_set_exact_products": (i, [i]),
    "vt_conv3x3_rows_colstats": (i, [p, i, i, i, p, i, i, p, p, p, p, p, C.c_long, C.POINTER(C.c_int), p]),
    "vt_gemm_sum_bn_relu_pool": (i, [i, i, C.POINTER(VtGemmTerm), i, i, p, i, p, p, p, i, p, p, p, p]),
    "vt_gemm_sum_colstats": (i, [i, i, C.POINTER(VtGemmTerm), i, i, f, p, i, C.c_long, p, i, p, p, C.c_long, C.POINTER(C.c_int), p]),
    "vt_gemm_bn_apply": (i, [i, i, i, p, i, p, p, p, p, p, i, i, p, p, i, p]),
    "vt_gemm_bn_apply_tile_records": (i, [i, i, i, p, i, p, p, p, p, p, i, i, p, p, i, p, p]),
    "vt_gemm_tn_records": (i, [i, i, i, p, i, C.c_long, p, C.c_long, p, p, i, C.c_long, i, p, C.c_long, p]),
    "vt_gemm_sum_records": (i, [i, C.POINTER(VtGemmTerm), p, i, p, i, i, p, i, C.c_long, p, i, p]),
    "vt_gemm_softmax_rows": (i, [i, i, p, i, i, i, i, f, p, i, C.c_long, p, i, C.c_long, p, C.c_long, i, p]),
    "vt_batchnorm_scratch_floats": (l, [i]),
    "vt_batchnorm_fwd": (i, [p, l, i, p, p, p, p, i, p, p, p, l, p]),
    "vt_batchnorm_bwd": (i, [p, p, p, p, l, i, p, p, p, p, l, p]),
    "vt_softmax_rows": (i, [p, l, i, p]),
    "vt_softmax_bwd_rows": (i, [p, p, l, i, p]),
    "vt_self_attention_fwd": (i, [p, p, p, p, p, p, p, i, i, i, i, p]),
    "vt_self_attention_bwd": (i, [p, p, p, p, p, p, p, p, p, p, p, p, p, p, l, i, i, i, i, p]),
    "vt_mha_fwd": (i, [p, p, p, p, p, p, l, i, i, i, i, i, p]),
    "vt_probe_copy": (i, [p, p, i, l, i, p]),
    "vt_sgd_update": (i, [p, p, i, f, f, i, p]),
    "vt_adam_update": (i, [p, p, p, p, i, f, f, f, f, i, f, p]),
    "vt_projector_tokens_bwd": (i, [p] * 8 + [l, p]),
    "vt_transformer_bwd_a": (i, [p] * 7 + [l, p]),
    "vt_projector_tokens_bwd_transformer_a": (i, [p] * 13 + [l, p]),
    "vt_transformer_bwd_b": (i, [p] * 14 + [l, p]),
    "vt_transformer_bwd_c": (i, [p] * 8 + [l, p]),
    "vt_mha_bwd": (i, [p] * 6 + [i, i, p]),
    "vt_mha_bwd_rows": (i, [p] * 4 + [i, i, p]),
    "vt_mha_bwd_wgrad": (i, [p] * 6 + [i, i, p]),
}
del p, i, l, f, u


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "vit_amd: %s is missing. Build it with `make -C vit_amd/csrc` (hipcc, gfx950); "
            "there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class VtError(RuntimeError):
    pass


def check(code, what=""):
    if code != 0:
        raise VtError("%s failed: %s (code %d)" % (what or "libvtcnn_hip", lib.vt_error_string(code).decode(), code))
