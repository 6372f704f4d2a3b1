"""ctypes binding of libchatllm_hip.so (include/chatllm_hip.h).  No fallbacks: errors raise."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("CLLM_LIB") or os.path.join(HERE, "libchatllm_hip.so")      # CLLM_LIB: A/B a differently built library (tools)
HEADER = os.path.join(os.path.dirname(HERE), "include", "chatllm_hip.h")


class CllmError(RuntimeError):
    pass


class CTensor(C.Structure):
    """struct cllm_tensor"""
    _fields_ = [("type", C.c_int32), ("ne", C.c_int64 * 4), ("nb", C.c_size_t * 4), ("data", C.c_void_p)]


class RopeParams(C.Structure):
    _fields_ = [("n_dims", C.c_int32), ("mode", C.c_int32), ("n_ctx_orig", C.c_int32), ("freq_base", C.c_float),
                ("freq_scale", C.c_float), ("ext_factor", C.c_float), ("attn_factor", C.c_float),
                ("beta_fast", C.c_float), ("beta_slow", C.c_float)]


class LlamaConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_layer", "hidden", "n_head", "n_kv_head", "head_dim", "ffn", "vocab", "max_len",
                                         "rope_mode")] + \
               [("rope_theta", C.c_float), ("rms_eps", C.c_float), ("qkv_bias", C.c_int32), ("tp_rank", C.c_int32),
                ("tp_size", C.c_int32), ("ffn_local", C.c_int32)]


ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)

# every exported symbol of include/chatllm_hip.h: name -> (restype, argtypes)
_P, _T = C.c_void_p, C.POINTER(CTensor)
SIGNATURES = {
    "cllm_abi_version": (C.c_int, []),
    "cllm_device_count": (C.c_int, []),
    "cllm_set_device": (C.c_int, [C.c_int]),
    "cllm_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "cllm_last_error": (C.c_char_p, []),
    "cllm_type_size": (C.c_size_t, [C.c_int]),
    "cllm_blck_size": (C.c_int, [C.c_int]),
    "cllm_row_size": (C.c_size_t, [C.c_int, C.c_int64]),
    "cllm_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "cllm_free": (C.c_int, [_P]),
    "cllm_memset": (C.c_int, [_P, C.c_int, C.c_size_t, _P]),
    "cllm_memcpy_h2d": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "cllm_memcpy_d2h": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "cllm_memcpy_d2d": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "cllm_host_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "cllm_host_free": (C.c_int, [_P]),
    "cllm_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "cllm_stream_destroy": (C.c_int, [_P]),
    "cllm_stream_sync": (C.c_int, [_P]),
    "cllm_scratch_release": (C.c_int, [_P]),
    "cllm_check_kernel_errors": (C.c_int, []),
    "cllm_graph_capture_begin": (C.c_int, [_P]),
    "cllm_graph_capture_end": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "cllm_graph_launch": (C.c_int, [_P, _P]),
    "cllm_graph_destroy": (C.c_int, [_P]),
    "cllm_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "cllm_event_destroy": (C.c_int, [_P]),
    "cllm_event_record": (C.c_int, [_P, _P]),
    "cllm_event_sync": (C.c_int, [_P]),
    "cllm_stream_wait_event": (C.c_int, [_P, _P]),
    "cllm_memcpy_peer_async": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_size_t, _P]),
    "cllm_event_elapsed_ms": (C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    "cllm_mul_mat_wsize": (C.c_size_t, [_T, _T]),
    "cllm_op_mul_mat": (C.c_int, [_P, _T, _T, _T, _P, C.c_size_t]),
    "cllm_mul_mat_ex_min_cols": (C.c_int, []),
    "cllm_set_prefill_mode": (C.c_int, [C.c_int]),
    "cllm_get_prefill_mode": (C.c_int, []),
    "cllm_set_prefill_attn_mode": (C.c_int, [C.c_int]),
    "cllm_op_mul_mat_ex": (C.c_int, [_P, _T, _T, _T, _P, C.c_size_t, C.c_int, _T, C.c_float, C.c_int, _T]),
    "cllm_set_decode_free_order": (C.c_int, [C.c_int]),
    "cllm_get_decode_free_order": (C.c_int, []),
    "cllm_ffn_fused_state_bytes": (C.c_size_t, [C.c_int64]),
    "cllm_op_ffn_fused": (C.c_int, [_P, _T, _T, _P, _P, C.c_float, _P, _P]),
    "cllm_bench_ffn": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64, _P, _P, C.c_float, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "cllm_op_mul_mat_vec_fused": (C.c_int, [_P, _T, C.c_int, _P, _P, C.c_float, C.c_int, _P, _P]),
    "cllm_pack_rows": (C.c_int, [_P, _P, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.c_size_t, C.c_int]),
    "cllm_bench_mul_mat_kernel": (C.c_int, [_P, _T, C.POINTER(C.c_void_p), C.c_int, _T, _T, _P, C.c_size_t, C.c_int, C.POINTER(C.c_float)]),
    "cllm_bench_mul_mat_id": (C.c_int, [_P, _T, _T, _T, C.POINTER(C.c_void_p), C.c_int, _T, _P, C.c_size_t, C.c_int, C.POINTER(C.c_float)]),
    "cllm_bench_read_bw": (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(C.c_float)]),
    "cllm_bench_gemv_fused": (C.c_int, [_P, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64, C.c_int, _P, _P, C.c_float, C.c_int, _P, _P, C.c_int,
                                        C.POINTER(C.c_float)]),
    "cllm_op_mul_mat_id": (C.c_int, [_P, _T, _T, _T, _T, _P, C.c_size_t]),
    "cllm_op_argmax_advance": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, C.c_int, _P]),
    "cllm_op_argmax_set": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, C.c_int, _P]),
    "cllm_op_snapshot_argmax_set": (C.c_int, [_P, _P, C.c_int, _P, C.c_int64, _P, _P, _P, C.c_int, _P]),
    "cllm_flash_attn_wsize": (C.c_size_t, [_T]),
    "cllm_op_flash_attn_ext": (C.c_int, [_P, _T, _T, _T, _T, _T, C.c_float, C.c_float, C.c_float, _P, C.c_size_t]),
    "cllm_attn_prefill_min_cols": (C.c_int, []),
    "cllm_op_attn_prefill": (C.c_int, [_P, _T, _T, _T, _T, C.c_float, C.c_int]),
    "cllm_op_mul_mat_id_silu_mul": (C.c_int, [_P, _T, _T, _T, _T]),
    "cllm_quantize_row_q8_0": (C.c_int, [_P, _P, _P, C.c_int64]),
    "cllm_quantize_row_q8_1": (C.c_int, [_P, _P, _P, C.c_int64]),
    "cllm_quantize_row_q8_K": (C.c_int, [_P, _P, _P, C.c_int64]),
    "cllm_vec_dot_isums": (C.c_int, [_P, C.c_int, C.c_int64, _P, _P, _P]),
    "cllm_op_rms_norm": (C.c_int, [_P, _T, _T, C.c_float]),
    "cllm_op_rms_norm_mul": (C.c_int, [_P, _T, _T, _T, C.c_float]),
    "cllm_op_rope": (C.c_int, [_P, _T, _T, _T, _T, C.POINTER(RopeParams)]),
    "cllm_op_soft_max": (C.c_int, [_P, _T, _T, _T, C.c_float, C.c_float]),
    "cllm_op_diag_mask_inf": (C.c_int, [_P, _T, _T, C.c_int]),
    "cllm_op_scale": (C.c_int, [_P, _T, _T, C.c_float, C.c_float]),
    "cllm_op_scale_mask_soft_max": (C.c_int, [_P, _T, _T, C.c_float, C.c_int]),
    "cllm_op_attn_decode": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int64, _P]),
    "cllm_op_rope_table": (C.c_int, [_P, _P, C.c_int, C.c_float, _P]),
    "cllm_attn_decode_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64]),
    "cllm_attn_decode_wsize": (C.c_size_t, [C.c_int64, C.c_int, C.c_int64]),
    "cllm_op_rope_kv_attn_decode": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int64, _P, _P, C.c_size_t]),
    "cllm_op_unary": (C.c_int, [_P, C.c_int, _T, _T]),
    "cllm_op_add": (C.c_int, [_P, _T, _T, _T]),
    "cllm_op_mul": (C.c_int, [_P, _T, _T, _T]),
    "cllm_op_div": (C.c_int, [_P, _T, _T, _T]),
    "cllm_op_sum_rows": (C.c_int, [_P, _T, _T]),
    "cllm_op_top_k": (C.c_int, [_P, _T, _T]),
    "cllm_op_moe_combine": (C.c_int, [_P, _T, _T, _T, _T, _T]),
    "cllm_op_mul_mat_id_combine": (C.c_int, [_P, _T, _T, _T, _T, _T, _T]),
    "cllm_op_moe_router": (C.c_int, [_P, _T, _T, C.c_float, _T, _T, _T, _T]),
    "cllm_op_moe_router_gate_up": (C.c_int, [_P, _T, _T, C.c_float, _T, _T, _T, _T, _T]),
    "cllm_op_silu_mul": (C.c_int, [_P, _T, _T, _T]),
    "cllm_op_set_rows": (C.c_int, [_P, _T, _T, _T]),
    "cllm_op_cpy": (C.c_int, [_P, _T, _T]),
    "cllm_op_get_rows": (C.c_int, [_P, _T, _T, _T]),
    "cllm_op_quantize_rows": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64, C.c_int64]),
    "cllm_dequantize_row": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64]),
    "cllm_llama_create": (C.c_int, [C.POINTER(LlamaConfig), _P, C.POINTER(C.c_void_p)]),
    "cllm_tp_split": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "cllm_llama_destroy": (None, [_P]),
    "cllm_llama_set_weight": (C.c_int, [_P, C.c_char_p, C.c_int, _P, C.c_size_t]),
    "cllm_llama_bind_weight": (C.c_int, [_P, C.c_char_p, C.c_int, _P, C.c_size_t]),
    "cllm_llama_set_allreduce": (C.c_int, [_P, ALLREDUCE_FN, _P]),
    "cllm_tp_unique_id": (C.c_int, [_P]),
    "cllm_tp_init": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cllm_tp_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cllm_tp_destroy": (C.c_int, [_P]),
    "cllm_tp_all_reduce_f32": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "cllm_llama_set_tp_comm": (C.c_int, [_P, _P]),
    "cllm_tp_oneshot_create": (C.c_int, [C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p), _P]),
    "cllm_tp_oneshot_connect": (C.c_int, [_P, _P]),
    "cllm_tp_oneshot_all_reduce_f32": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "cllm_tp_oneshot_error": (C.c_int, [_P]),
    "cllm_tp_oneshot_fine_grained": (C.c_int, [_P]),
    "cllm_tp_oneshot_clear_error": (C.c_int, [_P]),
    "cllm_tp_oneshot_destroy": (C.c_int, [_P]),
    "cllm_llama_set_tp_oneshot": (C.c_int, [_P, _P]),
    "cllm_tp_fused_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p), _P]),
    "cllm_tp_fused_connect": (C.c_int, [_P, _P]),
    "cllm_tp_fused_dev": (C.c_void_p, [_P]),
    "cllm_tp_fused_sites": (C.c_int, [_P]),
    "cllm_tp_fused_max_n": (C.c_size_t, [_P]),
    "cllm_tp_fused_fine_grained": (C.c_int, [_P]),
    "cllm_tp_fused_advance": (C.c_int, [_P, _P]),
    "cllm_tp_fused_error": (C.c_int, [_P]),
    "cllm_tp_fused_destroy": (C.c_int, [_P]),
    "cllm_tp_fused_create_group": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "cllm_tp_fused_clear_error": (C.c_int, [_P]),
    "cllm_op_mul_mat_vec_tp_scatter": (C.c_int, [_P, _T, C.c_int, _P, _P, C.c_int]),
    "cllm_op_mul_mat_vec_tp_gather": (C.c_int, [_P, _T, _P, _P, C.c_float, C.c_int, _P, _P, _P, C.c_int, _P]),
    "cllm_op_tp_gather_residual": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int, _P]),
    "cllm_op_kv_shard_copy": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int]),
    "cllm_copy_2d": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, C.c_size_t, C.c_size_t]),
    "cllm_llama_set_tp_fused": (C.c_int, [_P, _P]),
    "cllm_llama_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "cllm_llama_decode_greedy": (C.c_int, [_P, C.c_int32, C.c_int, C.c_int, _P]),
    "cllm_llama_decode_fused_logits": (C.c_int, [_P, C.c_int32, C.c_int, _P]),
    "cllm_llama_debug_read": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "cllm_llama_use_graph": (C.c_int, [_P, C.c_int]),
    "cllm_llama_weight_bytes": (C.c_size_t, [_P]),
}


def build(force=False):
    """compile every HIP source for gfx950 (hipcc cross-compiles without a GPU) -> libchatllm_hip.so"""
    src = os.path.join(HERE, "csrc")
    if not os.path.isdir(src):
        raise CllmError("csrc/ missing")
    args = ["make", "-C", src, "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        subprocess.check_call(["make", "-C", src, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return SO_PATH


_lib = None


def get():
    """the loaded library with prototypes applied; raises if the .so is absent (no silent fallback)"""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise CllmError(f"{SO_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            if "CLLM_LIB" in os.environ and not hasattr(lib, name):
                continue                     # A/B runs against an older build: newer entry points are simply absent
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = get().cllm_last_error()
        raise CllmError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def require_gpu():
    n = get().cllm_device_count()
    if n <= 0:
        raise CllmError("no HIP device visible: the gfx950 kernel library has no CPU fallback")
    return n
