"""ctypes binding of include/tsasr_hip.h  (the only door from Python into the HIP kernels).

The library is loaded lazily so that host-only logic (YAML loader, Brain bookkeeping, gradient
bucketing) can be imported and unit-tested in a container without a GPU; any attempt to *compute*
without the library raises ``TsasrHipMissing`` - there is no fallback path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSASR_HIP_LIB") or os.path.join(_HERE, "lib", "libtsasr_hip.so")   # (TSASR_HIP_LIB: A/B builds of the same ABI, tools only)

F32, BF16 = 0, 1


class TsasrHipMissing(RuntimeError):
    pass


class TsasrHipError(RuntimeError):
    pass


_lib = None

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
c_ll, c_ull = ctypes.c_longlong, ctypes.c_ulonglong

# name -> (restype, argtypes); mirrors include/tsasr_hip.h one to one
_PROTOS = {
    "tsasr_last_error": (ctypes.c_char_p, []),
    "tsasr_version": (c_int, []),
    "tsasr_device_ok": (c_int, []),
    "tsasr_joint_fwd": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_float, c_void_p]),
    "tsasr_joint_f32_fwd": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float, c_void_p]),
    "tsasr_joint_f32_bwd_workspace_bytes": (c_size_t, [c_int] * 3),
    "tsasr_joint_f32_bwd": (c_int, [c_void_p] * 10 + [c_int] * 6 + [c_float, c_void_p, c_size_t, c_void_p]),
    "tsasr_joint_bwd_workspace_bytes": (c_size_t, [c_int] * 4),
    "tsasr_joint_bwd": (c_int, [c_void_p] * 10 + [c_int] * 7 + [c_float, c_void_p, c_size_t, c_void_p]),
    "tsasr_rnnt_loss_workspace_bytes": (c_size_t, [c_int] * 3),
    "tsasr_rnnt_loss_error_word_offset": (ctypes.c_longlong, [c_int] * 3),
    "tsasr_rnnt_lattice_plan": (None, [c_int] * 3),
    "tsasr_linear_add_layernorm_ok": (c_int, [c_ll, c_int, c_int, c_ll, c_ll]),
    "tsasr_linear_add_layernorm_fwd": (c_int, [c_void_p, c_ll, c_void_p, c_ll] + [c_void_p] * 8 + [c_ll, c_int, c_int, c_float, c_float,
                                               c_ull, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "tsasr_rnnt_loss_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]),
    "tsasr_rnnt_loss_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]),
    "tsasr_layernorm_fwd": (c_int, [c_void_p] * 6 + [c_ll, c_int, c_float, c_float, c_int, c_void_p]),
    "tsasr_layernorm_bwd_workspace_bytes": (c_size_t, [c_ll, c_int]),
    "tsasr_layernorm_bwd": (c_int, [c_void_p] * 9 + [c_ll, c_int, c_float, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_layernorm_bwd_add": (c_int, [c_void_p] * 10 + [c_ll, c_int, c_float, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_seed_advance": (c_int, [c_void_p, c_ull, c_void_p]),
    "tsasr_bias_act_dropout_fwd": (c_int, [c_void_p] * 3 + [c_ll, c_int, c_float, c_float, c_ull, c_void_p, c_int, c_void_p]),
    "tsasr_colpart_workspace_bytes": (c_size_t, [c_ll, c_int]),
    "tsasr_bias_act_dropout_bwd": (c_int, [c_void_p] * 4 + [c_ll, c_int, c_float, c_float, c_ull, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_dropout_add_fwd": (c_int, [c_void_p] * 4 + [c_ll, c_int, c_float, c_float, c_ull, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "tsasr_dropout_add_bwd": (c_int, [c_void_p] * 3 + [c_ll, c_int, c_float, c_float, c_ull, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_dropout_add2_fwd": (c_int, [c_void_p] * 4 + [c_ll, c_int, c_float, c_float, c_ull, c_float, c_ull, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "tsasr_dropout_add2_bwd": (c_int, [c_void_p] * 4 + [c_ll, c_int, c_float, c_float, c_ull, c_float, c_ull, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_colsum_workspace_bytes": (c_size_t, [c_ll, c_int]),
    "tsasr_colsum": (c_int, [c_void_p, c_void_p, c_ll, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_convmod_fwd": (c_int, [c_void_p] * 10 + [c_int] * 5 + [c_float, c_float, c_int, c_void_p]),
    "tsasr_convmod_bwd_workspace_bytes": (c_size_t, [c_int] * 4),
    "tsasr_convmod_bwd": (c_int, [c_void_p] * 11 + [c_int] * 5 + [c_float, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_frontend_out_len": (c_int, [c_int]),
    "tsasr_frontend_c1_fwd": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_void_p]),
    "tsasr_frontend_c1_bwd_workspace_bytes": (c_size_t, [c_int]),
    "tsasr_frontend_c1_bwd": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]),
    "tsasr_conv3x3s2_fwd": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_void_p]),
    "tsasr_conv3x3s2_wgrad_workspace_bytes": (c_size_t, [c_int] * 4),
    "tsasr_conv3x3s2_wgrad": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]),
    "tsasr_conv3x3s2_wgrad_filters": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]),
    "tsasr_conv3x3s2_dgrad_plan_bytes": (c_size_t, [c_int] * 4),
    "tsasr_conv3x3s2_dgrad_plan": (c_int, [c_int] * 4 + [c_void_p, c_size_t]),
    "tsasr_conv3x3s2_dgrad": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]),
    "tsasr_frontend_im2col": (c_int, [c_void_p] * 2 + [c_int] * 6 + [c_void_p]),
    "tsasr_frontend_col2im": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    "tsasr_frontend_block_supported": (c_int, [c_int, c_int]),
    "tsasr_frontend_block_fwd": (c_int, [c_void_p] * 13 + [c_int] * 5 + [c_float, c_float, c_float, c_ull, c_float, c_ull, c_void_p, c_int, c_void_p]),
    "tsasr_frontend_block_dparams": (c_size_t, [c_int] * 3),
    "tsasr_frontend_block_bwd_workspace_bytes": (c_size_t, [c_int] * 3),
    "tsasr_frontend_block_bwd": (c_int, [c_void_p] * 15 + [c_int] * 5 + [c_float, c_float, c_ull, c_float, c_ull, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_relpos_attn_lds_bytes": (c_size_t, []),
    "tsasr_relpos_attn_fwd": (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_float, c_int, c_float, c_ull, c_void_p, c_int, c_void_p]),
    "tsasr_relpos_attn_fwd_workspace_bytes": (c_size_t, [c_int] * 3),
    "tsasr_relpos_attn_fwd_ws": (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_float, c_int, c_float, c_ull, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_relpos_attn_bwd_workspace_bytes": (c_size_t, [c_int] * 3),
    "tsasr_relpos_attn_bwd": (c_int, [c_void_p] * 12 + [c_int] * 4 + [c_float, c_int, c_float, c_ull, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_relpos_dpk_defer": (c_int, [c_int]),
    "tsasr_relpos_dpk_pending": (c_int, []),
    "tsasr_relpos_dpk_table_bytes": (c_size_t, [c_int]),
    "tsasr_relpos_dpk_flush": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "tsasr_relpos_dpk_discard": (None, []),
    "tsasr_relpos_attn_keepbits_bytes": (c_size_t, [c_int] * 3),
    "tsasr_relpos_attn_keepbits": (None, [c_void_p]),
    "tsasr_accumulate_many": (c_int, [c_void_p, c_int, c_void_p]),
    "tsasr_clip_adamw_workspace_bytes": (c_size_t, []),
    "tsasr_clip_adamw_step": (c_int, [c_void_p] * 8 + [c_ll] + [c_float] * 5 + [c_void_p, c_size_t, c_void_p]),
    "tsasr_gemm_set_plan": (None, [c_int, c_int]),
    "tsasr_gemm_set_ring": (None, [c_int]),
    "tsasr_gemm_set_lab_floor": (None, [c_int]),
    "tsasr_gemm_bf16_workspace_bytes": (c_size_t, [c_int] * 4),
    "tsasr_gemm_f32": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_ll] * 3 + [c_int] * 3 + [c_void_p]),
    "tsasr_gemm_bf16_nt_batched": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_ll] * 4 + [c_int, c_void_p]),
    "tsasr_gemm_bf16_fused_workspace_bytes": (c_size_t, [c_int] * 2),
    "tsasr_gemm_bf16_fused": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_ll] * 3 + [c_int] * 3 + [c_void_p, c_void_p, c_ll, c_float, c_float, c_ull, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tsasr_gemm_bf16_fused_mask_ok": (c_int, [c_int] * 3),
    "tsasr_gemm_bf16": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_ll] * 3 + [c_int] * 4 + [c_void_p, c_size_t, c_void_p]),
    "tsasr_lstm_step_fwd": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "tsasr_lstm_step_bwd": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "tsasr_lstm_cell_fwd": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "tsasr_lstm_cell_bwd": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "tsasr_fbank_workspace_bytes": (c_size_t, [c_int] * 3),
    "tsasr_fbank_fwd": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_float, c_float, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_sentence_norm_fwd": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_float, c_int, c_int, c_void_p]),
    "tsasr_add_layernorm_fwd": (c_int, [c_void_p] * 9 + [c_ll, c_int, c_float, c_float, c_ull, c_void_p, c_void_p, c_int, c_float, c_int, c_void_p]),
    "tsasr_add_layernorm_bwd_workspace_bytes": (c_size_t, [c_ll, c_int]),
    "tsasr_add_layernorm_bwd": (c_int, [c_void_p] * 11 + [c_ll, c_int, c_float, c_float, c_ull, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_add_layernorm2_fwd": (c_int, [c_void_p] * 14 + [c_ll, c_int, c_float, c_float, c_ull, c_void_p, c_void_p, c_int, c_float, c_float, c_int, c_void_p]),
    "tsasr_add_layernorm2_bwd_workspace_bytes": (c_size_t, [c_ll, c_int]),
    "tsasr_add_layernorm2_bwd": (c_int, [c_void_p] * 18 + [c_ll, c_int, c_float, c_float, c_ull, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_lstm_onehot_gates": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p]),
    "tsasr_lstm_seq_persistent": (c_int, [c_int, c_int, c_int]),
    "tsasr_lstm_seq_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "tsasr_lstm_seq_fwd": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p, c_size_t, c_void_p]),
    "tsasr_lstm_seq_bwd": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p, c_size_t, c_void_p]),
    "tsasr_transpose_many_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "tsasr_reduce_defer": (c_int, [c_int]),
    "tsasr_reduce_pending": (c_int, []),
    "tsasr_reduce_discard": (None, []),
    "tsasr_reduce_table_bytes": (c_size_t, [c_int]),
    "tsasr_reduce_flush": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "tsasr_reduce_flush_stream": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "tsasr_mean_pool_fwd": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "tsasr_mean_pool_bwd": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "tsasr_abs_lengths": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "tsasr_inject_fwd": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "tsasr_inject_bwd": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "tsasr_greedy_decode": (c_int, [c_void_p] * 12 + [c_int] * 7 + [c_float, c_int, c_int, c_void_p]),
    "tsasr_count_nonfinite": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "tsasr_allreduce_load": (c_int, [ctypes.c_char_p]),
    "tsasr_allreduce_unique_id": (c_int, [c_void_p]),
    "tsasr_allreduce_init": (c_int, [c_void_p, c_int, c_int]),
    "tsasr_allreduce_ready": (c_int, []),
    "tsasr_allreduce_bucket": (c_int, [c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "tsasr_allreduce_destroy": (c_int, []),
    "tsasr_wgrad_queue": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_ll] * 3),
    "tsasr_wgrad_pending": (c_int, []),
    "tsasr_wgrad_table_bytes": (c_size_t, [c_int]),
    "tsasr_wgrad_flush": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "tsasr_wgrad_discard": (None, []),
    "tsasr_wgrad_next_flush_slots": (None, [c_int]),
    "tsasr_wgrad_next_flush_wgs": (None, [c_int]),
    "tsasr_specaug_params_words": (c_size_t, [c_int] * 3),
    "tsasr_specaug_draw": (c_int, [c_void_p] + [c_int] * 10 + [c_ull, c_void_p, c_void_p]),
    "tsasr_specaug_workspace_bytes": (c_size_t, []),
    "tsasr_specaug_apply": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p, c_size_t, c_void_p]),
    "tsasr_resample_out_len": (c_ll, [c_ll, c_int, c_int]),
    "tsasr_resample_fwd": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "tsasr_lstm_f32_fwd": (c_int, [c_void_p] * 12 + [c_int] * 4 + [c_void_p]),
    "tsasr_lstm_f32_bwd": (c_int, [c_void_p] * 10 + [c_int] * 3 + [c_void_p]),
    "tsasr_attn_f32_fwd": (c_int, [c_void_p] * 10 + [c_int] * 5 + [c_float, c_int, c_float, c_ull, c_void_p, c_int, c_void_p]),
    "tsasr_attn_f32_bwd_workspace_bytes": (c_size_t, [c_int] * 5),
    "tsasr_attn_f32_bwd": (c_int, [c_void_p] * 18 + [c_int] * 5 + [c_float, c_int, c_float, c_ull, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "tsasr_mix_sources_workspace_bytes": (c_size_t, []),
    "tsasr_mix_sources_out_len": (c_ll, [c_void_p, c_void_p, c_int, c_ll, c_ll]),
    "tsasr_mix_sources": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_ll, c_ll, c_void_p, c_void_p, c_size_t, c_void_p]),
}


def exported_symbols():
    return sorted(_PROTOS)


def lib():
    """The loaded shared library; raises TsasrHipMissing when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TsasrHipMissing(
                f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or __graft_entry__.build()). ts-asr_amd has no CPU/ATen fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)  # AttributeError = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_lab = None
LAB_PATH = os.path.join(_HERE, "lib", "libtsasr_lab.so")


def lab():
    """Lab equipment (include/tsasr_lab.h: LDS / memory fills, wall-clock stamp) for tests/helpers, tools and prof's TSASR_STAMPS mode -
    a separate library; the product path never loads it."""
    global _lab
    if _lab is None:
        if not os.path.exists(LAB_PATH):
            raise TsasrHipMissing(f"{LAB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}`")
        L = ctypes.CDLL(LAB_PATH)
        L.tsasr_lab_fill_lds.argtypes, L.tsasr_lab_fill_lds.restype = [ctypes.c_uint, c_void_p], c_int
        L.tsasr_lab_fill.argtypes, L.tsasr_lab_fill.restype = [c_void_p, ctypes.c_uint, c_size_t, c_void_p], c_int
        L.tsasr_lab_stamp.argtypes, L.tsasr_lab_stamp.restype = [c_void_p, c_void_p], c_int
        _lab = L
    return _lab


def check(rc, what=""):
    if rc != 0:
        msg = lib().tsasr_last_error().decode(errors="replace")
        raise TsasrHipError(f"{what} failed (rc={rc}): {msg}")


def io_dtype(t):
    import torch

    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"ts-asr_amd kernels take float32 or bfloat16 activations, got {t.dtype}")


def require_gpu(*tensors):
    """Product ops never run on the CPU."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise TsasrHipMissing("ts-asr_amd ops need tensors on an MI355X (cuda) device; there is no CPU path "
                                  "(the CPU restatement lives in oracle/ and is test infrastructure only)")


def stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())
