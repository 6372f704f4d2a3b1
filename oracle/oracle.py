"""ctypes/numpy front-end of oracle/liboracle.so (the C restatement) and oracle/_ref/libref_ops.so
(the real reference, when built).  TEST INFRASTRUCTURE ONLY: imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

F32, F16, Q4_0, Q4_1, Q8_0, Q8_1, Q4_K, Q5_K, Q6_K, Q8_K, I32, I64 = 0, 1, 2, 3, 8, 9, 12, 13, 14, 15, 26, 27
Q5_0, Q5_1, Q2_K, Q3_K, IQ4_NL, MXFP4, IQ4_XS, TQ1_0, TQ2_0, IQ2_XXS, IQ2_XS, IQ3_XXS, IQ3_S, IQ2_S, IQ1_S, IQ1_M = 6, 7, 10, 11, 20, 39, 23, 34, 35, 16, 17, 18, 21, 22, 19, 29
TYPE_SIZE = {F32: 4, F16: 2, Q4_0: 18, Q4_1: 20, Q8_0: 34, Q8_1: 36, Q4_K: 144, Q5_K: 176, Q6_K: 210, Q8_K: 292, I32: 4, I64: 8,
             Q5_0: 22, Q5_1: 24, Q2_K: 84, Q3_K: 110, IQ4_NL: 18, MXFP4: 17, IQ4_XS: 136, TQ1_0: 54, TQ2_0: 66, IQ2_XXS: 66, IQ2_XS: 74, IQ2_S: 82, IQ3_XXS: 98, IQ3_S: 110, IQ1_S: 50, IQ1_M: 56}
BLCK = {F32: 1, F16: 1, Q4_0: 32, Q4_1: 32, Q8_0: 32, Q8_1: 32, Q4_K: 256, Q5_K: 256, Q6_K: 256, Q8_K: 256, I32: 1, I64: 1,
        Q5_0: 32, Q5_1: 32, Q2_K: 256, Q3_K: 256, IQ4_NL: 32, MXFP4: 32, IQ4_XS: 256, TQ1_0: 256, TQ2_0: 256, IQ2_XXS: 256, IQ2_XS: 256, IQ2_S: 256, IQ3_XXS: 256, IQ3_S: 256, IQ1_S: 256, IQ1_M: 256}
NP_OF = {F32: np.float32, F16: np.float16, I32: np.int32, I64: np.int64}


def row_size(t, ne):
    return TYPE_SIZE[t] * (ne // BLCK[t])


class Tensor(C.Structure):
    _fields_ = [("type", C.c_int32), ("ne", C.c_int64 * 4), ("nb", C.c_size_t * 4), ("data", C.c_void_p)]


def build(force=False):
    so = os.path.join(HERE, "liboracle.so")
    src = [os.path.join(HERE, f) for f in ("ggml_oracle.c", "ggml_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_fp16_to_fp32.restype = C.c_float
        _lib.orc_fp16_to_fp32.argtypes = [C.c_uint16]
        _lib.orc_fp32_to_fp16.restype = C.c_uint16
        _lib.orc_fp32_to_fp16.argtypes = [C.c_float]
        for n in ("orc_vec_dot_q4_0_q8_0", "orc_vec_dot_q8_0_q8_0", "orc_vec_dot_q4_1_q8_1", "orc_vec_dot_q4_K_q8_K"):
            getattr(_lib, n).restype = C.c_float
            getattr(_lib, n).argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_expf_avx2.restype = C.c_float
        _lib.orc_expf_avx2.argtypes = [C.c_float]
        _lib.orc_silu_avx2.restype = C.c_float
        _lib.orc_silu_avx2.argtypes = [C.c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def tensor(arr, type_, ne, nb=None, offset=0):
    """describe (a view of) a numpy buffer as a strided 4-D tensor; nb in bytes (default: dense)"""
    ne = list(ne) + [1] * (4 - len(ne))
    if nb is None:
        nb = [TYPE_SIZE[type_], row_size(type_, ne[0])]
        nb.append(nb[1] * ne[1])
        nb.append(nb[2] * ne[2])
    nb = list(nb) + [nb[-1]] * (4 - len(nb))
    t = Tensor()
    t.type = type_
    t.ne[:] = ne
    t.nb[:] = nb
    t.data = arr.ctypes.data + offset
    t._keep = arr
    return t


# ---- quantizers / dequantizers -------------------------------------------------------------
def quantize_q8_0(x, ref=False):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros(x.size // 32 * 34, np.uint8)
    (lib().orc_quantize_row_q8_0_ref if ref else lib().orc_quantize_row_q8_0)(_p(x), _p(y), C.c_int64(x.size))
    return y


def quantize_q8_1(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros(x.size // 32 * 36, np.uint8)
    lib().orc_quantize_row_q8_1(_p(x), _p(y), C.c_int64(x.size))
    return y


def quantize_q8_K(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros(x.size // 256 * 292, np.uint8)
    lib().orc_quantize_row_q8_K(_p(x), _p(y), C.c_int64(x.size))
    return y


def quantize_ref(type_, x):
    """the reference's from_float_ref (what chatllm.cpp's on-load re-quantization calls) restated: bytes of the quantized row"""
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros(TYPE_SIZE[type_] * (x.size // BLCK[type_]), np.uint8)
    _chk(lib().orc_quantize_row_ref(C.c_int(type_), _p(x), _p(y), C.c_int64(x.size)), "quantize_row_ref")
    return y


def dequantize(type_, blocks, k):
    blocks = np.ascontiguousarray(blocks, np.uint8)
    y = np.zeros(k, np.float32)
    lib().orc_dequantize_row(C.c_int(type_), _p(blocks), _p(y), C.c_int64(k))
    return y


def vec_dot(wtype, n, w, a):
    """returns (float result, exact int32 block sums)"""
    fn = {Q4_0: lib().orc_vec_dot_q4_0_q8_0, Q8_0: lib().orc_vec_dot_q8_0_q8_0, Q4_1: lib().orc_vec_dot_q4_1_q8_1, Q4_K: lib().orc_vec_dot_q4_K_q8_K}[wtype]
    isums = np.zeros(n // 32 if wtype != Q4_K else 2 * (n // 256), np.int32)
    s = fn(C.c_int64(n), _p(np.ascontiguousarray(w)), _p(np.ascontiguousarray(a)), _p(isums))
    return np.float32(s), isums


# ---- ops on tensor descriptors --------------------------------------------------------------
def _chk(rc, name):
    if rc != 0:
        raise RuntimeError(f"oracle {name} failed rc={rc}")


def mul_mat(w, x, dst):
    _chk(lib().orc_mul_mat(C.byref(w), C.byref(x), C.byref(dst)), "mul_mat")


def mul_mat_id(w, x, ids, dst):
    _chk(lib().orc_mul_mat_id(C.byref(w), C.byref(x), C.byref(ids), C.byref(dst)), "mul_mat_id")


def rms_norm(src, dst, eps):
    _chk(lib().orc_rms_norm(C.byref(src), C.byref(dst), C.c_float(eps)), "rms_norm")


class RopeParams(C.Structure):
    _fields_ = [("n_dims", C.c_int32), ("mode", C.c_int32), ("n_ctx_orig", C.c_int32), ("freq_base", C.c_float),
                ("freq_scale", C.c_float), ("ext_factor", C.c_float), ("attn_factor", C.c_float),
                ("beta_fast", C.c_float), ("beta_slow", C.c_float)]


def rope(src, pos, ff, dst, n_dims, mode, freq_base, n_ctx_orig=0, freq_scale=1.0, ext_factor=0.0, attn_factor=1.0,
         beta_fast=0.0, beta_slow=0.0):
    p = RopeParams(n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow)
    pos = np.ascontiguousarray(pos, np.int32)
    ffp = _p(np.ascontiguousarray(ff, np.float32)) if ff is not None else None
    _chk(lib().orc_rope(C.byref(src), _p(pos), ffp, C.byref(dst), C.byref(p)), "rope")


def soft_max(src, mask, dst, scale=1.0):
    _chk(lib().orc_soft_max(C.byref(src), C.byref(mask) if mask is not None else None, C.byref(dst), C.c_float(scale),
                            C.c_float(0.0)), "soft_max")


def flash_attn_ext(q, k, v, mask, dst, scale):
    _chk(lib().orc_flash_attn_ext(C.byref(q), C.byref(k), C.byref(v), C.byref(mask) if mask is not None else None, C.byref(dst),
                                  C.c_float(scale)), "flash_attn_ext")


def diag_mask_inf(src, dst, n_past):
    _chk(lib().orc_diag_mask_inf(C.byref(src), C.byref(dst), C.c_int(n_past)), "diag_mask_inf")


def scale(src, dst, s, b=0.0):
    _chk(lib().orc_scale(C.byref(src), C.byref(dst), C.c_float(s), C.c_float(b)), "scale")


def silu(src, dst):
    _chk(lib().orc_silu(C.byref(src), C.byref(dst)), "silu")


def add(a, b, dst):
    _chk(lib().orc_add(C.byref(a), C.byref(b), C.byref(dst)), "add")


def mul(a, b, dst):
    _chk(lib().orc_mul(C.byref(a), C.byref(b), C.byref(dst)), "mul")


def div(a, b, dst):
    _chk(lib().orc_div(C.byref(a), C.byref(b), C.byref(dst)), "div")


def sum_rows(src, dst):
    _chk(lib().orc_sum_rows(C.byref(src), C.byref(dst)), "sum_rows")


def top_k(src, dst):
    _chk(lib().orc_top_k(C.byref(src), C.byref(dst)), "top_k")


def set_rows(src, idx, dst):
    _chk(lib().orc_set_rows(C.byref(src), C.byref(idx), C.byref(dst)), "set_rows")


def cpy(src, dst):
    _chk(lib().orc_cpy(C.byref(src), C.byref(dst)), "cpy")


def get_rows(src, idx, dst):
    _chk(lib().orc_get_rows(C.byref(src), C.byref(idx), C.byref(dst)), "get_rows")


# ---- whole model ----------------------------------------------------------------------------
class LlamaConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_layer", "hidden", "n_head", "n_kv_head", "head_dim", "ffn", "vocab", "max_len",
                                         "rope_mode")] + [("rope_theta", C.c_float), ("rms_eps", C.c_float), ("qkv_bias", C.c_int32)]


class Weight(C.Structure):
    _fields_ = [("type", C.c_int32), ("data", C.c_void_p)]


class LlamaLayer(C.Structure):
    _fields_ = [("attn_norm", C.c_void_p), ("ffn_norm", C.c_void_p)] + \
               [(n, Weight) for n in ("wq", "wk", "wv", "wo", "wgate", "wup", "wdown")] + \
               [("bq", C.c_void_p), ("bk", C.c_void_p), ("bv", C.c_void_p)]


class LlamaModel(C.Structure):
    _fields_ = [("cfg", LlamaConfig), ("tok_embd", Weight), ("lm_head", Weight), ("out_norm", C.c_void_p),
                ("layers", C.POINTER(LlamaLayer)), ("k_cache", C.c_void_p), ("v_cache", C.c_void_p)]


class Llama:
    """CPU restatement of a Llama-3/Qwen2-style decoder over a dict of numpy weights (see chatllm.cpp_amd synth)."""

    def __init__(self, cfg, weights):
        self.cfg, self.w = cfg, weights
        c = LlamaConfig(cfg["n_layer"], cfg["hidden"], cfg["n_head"], cfg["n_kv_head"], cfg["head_dim"], cfg["ffn"],
                        cfg["vocab"], cfg["max_len"], cfg.get("rope_mode", 0), cfg.get("rope_theta", 500000.0),
                        cfg.get("rms_eps", 1e-5), 1 if cfg.get("qkv_bias") else 0)
        self._layers = (LlamaLayer * cfg["n_layer"])()

        def W(name):
            t, a = weights[name]
            return Weight(t, a.ctypes.data)

        def P(name):
            return weights[name][1].ctypes.data if name in weights else None

        for i in range(cfg["n_layer"]):
            L, p = self._layers[i], f"layers.{i}."
            L.attn_norm, L.ffn_norm = P(p + "attn_norm"), P(p + "ffn_norm")
            for n in ("wq", "wk", "wv", "wo", "wgate", "wup", "wdown"):
                setattr(L, n, W(p + n))
            L.bq, L.bk, L.bv = P(p + "bq"), P(p + "bk"), P(p + "bv")
        kd = cfg["n_kv_head"] * cfg["head_dim"]
        self.k_cache = np.zeros((cfg["n_layer"], cfg["max_len"], kd), np.uint16)
        self.v_cache = np.zeros((cfg["n_layer"], kd, cfg["max_len"]), np.uint16)
        self.m = LlamaModel(c, W("tok_embd"), W("lm_head"), P("out_norm"), self._layers, self.k_cache.ctypes.data,
                            self.v_cache.ctypes.data)
        self.n_past = 0

    def forward(self, tokens):
        tokens = np.ascontiguousarray(tokens, np.int32)
        logits = np.zeros(self.cfg["vocab"], np.float32)
        _chk(lib().orc_llama_forward(C.byref(self.m), _p(tokens), C.c_int(tokens.size), C.c_int(self.n_past), _p(logits)),
             "llama_forward")
        self.n_past += tokens.size
        return logits


# ---- the real reference (oracle/_ref), optional ----------------------------------------------
_ref = None


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libref_ops.so"))


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.path.join(HERE, "_ref", "libref_ops.so"))
        _ref.ref_row_size.restype = C.c_size_t
        _ref.ref_set_threads(C.c_int(min(8, os.cpu_count() or 1)))
    return _ref
