"""Operator surface of the hot path, named after the reference's `chatllm::ggml::*` wrappers
(src/layers.h:41-304; bodies src/layers.cpp:602-614, 893-983, 1062-1107).  Each function
allocates its result like the ggml op constructor would and launches the HIP kernel through
the C ABI.  Everything runs on the default stream; results are device `Tensor`s.
"""
import ctypes as C

from . import lib as _l
from .tensor import Tensor, F32, F16, I32, I64, Buffer  # noqa: F401

ROPE_NORMAL, ROPE_NEOX = 0, 2
UNARY_SILU = 10

_wdata = {"buf": None}


def _scratch(nbytes):
    b = _wdata["buf"]
    if b is None or b.nbytes < nbytes:
        _wdata["buf"] = b = Buffer(max(nbytes, 1 << 20))
    return b


def _ref(t):
    return C.byref(t.c()) if t is not None else None


def mul_mat(a, b, dst=None):
    """ggml::mul_mat(ctx, a, b): dst[ne01, ne11, ne12, ne13] = a^T . b"""
    L = _l.get()
    if dst is None:
        dst = Tensor(F32, [a.ne[1], b.ne[1], b.ne[2], b.ne[3]])
    ca, cb, cd = a.c(), b.c(), dst.c()
    ws = L.cllm_mul_mat_wsize(C.byref(ca), C.byref(cb))
    buf = _scratch(ws) if ws else None
    _l.check(L.cllm_op_mul_mat(None, C.byref(ca), C.byref(cb), C.byref(cd), buf.ptr if buf else None, buf.nbytes if buf else 0), "mul_mat")
    return dst


def mul_mat_ex(a, b, pro=0, norm_w=None, eps=0.0, epi=0, resid=None, dst=None):
    """the prefill form of the node patterns around a quantized MUL_MAT (cllm_op_mul_mat_ex): norm / SiLU*up prologue, SiLU*up / residual epilogue"""
    L = _l.get()
    K = a.ne[0]
    if dst is None:
        dst = Tensor(F32, [a.ne[1] // 2 if epi else a.ne[1], b.ne[1]])
    geom = Tensor(F32, [K, b.ne[1]]) if pro == 3 else b                      # (only its shape is used: the scratch size)
    ws = L.cllm_mul_mat_wsize(_ref(a), _ref(geom))
    buf = _scratch(ws)
    _l.check(L.cllm_op_mul_mat_ex(None, _ref(a), _ref(b), _ref(dst), buf.ptr, buf.nbytes, pro, _ref(norm_w), float(eps), epi, _ref(resid)), "mul_mat_ex")
    return dst


def mul_mat_id(as_, b, ids, dst=None):
    """ggml::mul_mat_id(ctx, as, b, ids)"""
    L = _l.get()
    if dst is None:
        dst = Tensor(F32, [as_.ne[1], ids.ne[0], b.ne[2], 1])
    ca, cb, ci, cd = as_.c(), b.c(), ids.c(), dst.c()
    ws = L.cllm_mul_mat_wsize(C.byref(ca), C.byref(cb))
    buf = _scratch(ws)
    _l.check(L.cllm_op_mul_mat_id(None, C.byref(ca), C.byref(cb), C.byref(ci), C.byref(cd), buf.ptr, buf.nbytes), "mul_mat_id")
    return dst


def rms_norm(a, eps, dst=None):
    dst = dst or Tensor(F32, a.ne)
    _l.check(_l.get().cllm_op_rms_norm(None, _ref(a), _ref(dst), eps), "rms_norm")
    return dst


def rms_norm_mul(a, weight, eps, dst=None):
    """RMSNorm::forward: rms_norm then mul by the weight vector"""
    dst = dst or Tensor(F32, a.ne)
    _l.check(_l.get().cllm_op_rms_norm_mul(None, _ref(a), _ref(weight), _ref(dst), eps), "rms_norm_mul")
    return dst


def rope_ext(a, pos, freq_factors, n_dims, mode, n_ctx_orig=0, freq_base=10000.0, freq_scale=1.0, ext_factor=0.0,
             attn_factor=1.0, beta_fast=0.0, beta_slow=0.0, inplace=False):
    """ggml::rope_ext / rope_ext_inplace"""
    dst = a if inplace else Tensor(F32, a.ne)
    p = _l.RopeParams(n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow)
    _l.check(_l.get().cllm_op_rope(None, _ref(a), _ref(pos), _ref(freq_factors), _ref(dst), C.byref(p)), "rope")
    return dst


def soft_max(a, dst=None):
    return soft_max_ext(a, None, 1.0, 0.0, dst)


def soft_max_ext(a, mask, scale, max_bias, dst=None):
    dst = dst or Tensor(F32, a.ne)
    _l.check(_l.get().cllm_op_soft_max(None, _ref(a), _ref(mask), _ref(dst), scale, max_bias), "soft_max")
    return dst


def diag_mask_inf(a, n_past, dst=None):
    dst = dst or Tensor(F32, a.ne)
    _l.check(_l.get().cllm_op_diag_mask_inf(None, _ref(a), _ref(dst), n_past), "diag_mask_inf")
    return dst


def scale(a, s, b=0.0, dst=None):
    dst = dst or Tensor(F32, a.ne)
    _l.check(_l.get().cllm_op_scale(None, _ref(a), _ref(dst), s, b), "scale")
    return dst


def scale_mask_soft_max(a, scale_, n_past, dst=None):
    dst = dst or Tensor(F32, a.ne)
    _l.check(_l.get().cllm_op_scale_mask_soft_max(None, _ref(a), _ref(dst), scale_, n_past), "scale_mask_soft_max")
    return dst


def flash_attention(q, k, v, mask, scale, max_bias=0.0, logit_softcap=0.0, dst=None):
    """ggml::flash_attention(ctx, q, k, v, mask, scale) (src/layers.cpp:1506-1518): q [D, N, H, B] F32; k, v [D, n_kv, Hkv, B] F16 | Q8_0;
    mask F16 [n_kv, N] or None -> [D, H, N, B] F32"""
    L = _l.get()
    dst = dst or Tensor(F32, [v.ne[0], q.ne[2], q.ne[1], q.ne[3]])
    cq = q.c()
    ws = L.cllm_flash_attn_wsize(C.byref(cq))
    buf = _scratch(ws)
    _l.check(L.cllm_op_flash_attn_ext(None, C.byref(cq), _ref(k), _ref(v), _ref(mask), _ref(dst), scale, max_bias, logit_softcap, buf.ptr, buf.nbytes), "flash_attn_ext")
    return dst


def attn_prefill(q, k, vt, scale, n_past, dst=None):
    """the eager prefill attention block as one flash kernel: q [D, N, H] F32, k [D, n_kv, Hkv] F16 rows, vt [n_kv, D, Hkv] F16 -> [D, N, H] F32"""
    dst = dst or Tensor(F32, [vt.ne[1], q.ne[1], q.ne[2], q.ne[3]])
    _l.check(_l.get().cllm_op_attn_prefill(None, _ref(q), _ref(k), _ref(vt), _ref(dst), scale, n_past), "attn_prefill")
    return dst


def attn_decode(q, pos, n_head, n_kv_head, head_dim, k_cache, v_cache, max_len, dst=None):
    """fused single-token attention; q: [hd, n_head] F32, pos: I32 [1] device tensor holding n_past"""
    dst = dst or Tensor(F32, [head_dim * n_head])
    _l.check(_l.get().cllm_op_attn_decode(None, q.data_ptr(), pos.data_ptr(), n_head, n_kv_head, head_dim, k_cache.data_ptr(), v_cache.data_ptr(),
                                          max_len, dst.data_ptr()), "attn_decode")
    return dst


def rope_kv_attn_decode(qkv, pos, n_kv, n_head, n_kv_head, head_dim, rope_mode, freq_base, k_cache, v_cache, max_len, table=True, dst=None):
    """KVCacheAttention for one token in one call: RoPE(q,k), K/V cache write at `pos`, scores, softmax, V.P (src/layers.cpp:3044-3123,
    2499-2561); qkv: the un-rotated projections, pos: I32 [1] device tensor == n_kv - 1"""
    L = _l.get()
    dst = dst or Tensor(F32, [head_dim * n_head])
    cs = None
    if table:
        cs = Tensor(F32, [head_dim])
        _l.check(L.cllm_op_rope_table(None, pos.data_ptr(), head_dim, freq_base, cs.data_ptr()), "rope_table")
    ws = L.cllm_attn_decode_wsize(n_kv, n_head, max_len)
    buf = _scratch(ws) if ws else None
    _l.check(L.cllm_op_rope_kv_attn_decode(None, qkv.data_ptr(), pos.data_ptr(), cs.data_ptr() if cs else None, freq_base, n_kv, n_head, n_kv_head, head_dim,
                                           rope_mode, k_cache.data_ptr(), v_cache.data_ptr(), max_len, dst.data_ptr(), buf.ptr if buf else None,
                                           buf.nbytes if buf else 0), "rope_kv_attn_decode")
    return dst


def silu(a, dst=None):
    dst = dst or Tensor(F32, a.ne)
    _l.check(_l.get().cllm_op_unary(None, UNARY_SILU, _ref(a), _ref(dst)), "silu")
    return dst


def add(a, b, dst=None):
    dst = dst or Tensor(F32, a.ne)
    _l.check(_l.get().cllm_op_add(None, _ref(a), _ref(b), _ref(dst)), "add")
    return dst


def mul(a, b, dst=None):
    dst = dst or Tensor(F32, a.ne)
    _l.check(_l.get().cllm_op_mul(None, _ref(a), _ref(b), _ref(dst)), "mul")
    return dst


def div(a, b, dst=None):
    dst = dst or Tensor(F32, a.ne)
    _l.check(_l.get().cllm_op_div(None, _ref(a), _ref(b), _ref(dst)), "div")
    return dst


def sum_rows(a, dst=None):
    dst = dst or Tensor(F32, [1, a.ne[1], a.ne[2], a.ne[3]])
    _l.check(_l.get().cllm_op_sum_rows(None, _ref(a), _ref(dst)), "sum_rows")
    return dst


def top_k(a, k, dst=None):
    """ggml::top_k: I32 [k, ...] indices of the k largest of each row (the reference's order: descending, first two swapped)"""
    dst = dst or Tensor(I32, [k, a.ne[1], a.ne[2], a.ne[3]])
    _l.check(_l.get().cllm_op_top_k(None, _ref(a), _ref(dst)), "top_k")
    return dst


def mul_mat_id_silu_mul(as_gate, as_up, b, ids, dst=None):
    """MultiMLP::forward's gate / up / SiLU / MUL for one token in one launch (the two expert tensors are packed first: rows alternating)"""
    L = _l.get()
    K, F, E = as_gate.ne[0], as_gate.ne[1], as_gate.ne[2]
    packed = Tensor(as_gate.type, [K, 2 * F, E])
    srcs = (C.c_void_p * 2)(as_gate.data_ptr().value, as_up.data_ptr().value)
    rows = (C.c_int64 * 2)(F * E, F * E)
    _l.check(L.cllm_pack_rows(None, packed.data_ptr(), srcs, rows, 2, as_gate.nb[1], 1), "pack_rows")
    dst = dst or Tensor(F32, [F, ids.ne[0], 1])
    _l.check(L.cllm_op_mul_mat_id_silu_mul(None, _ref(packed), _ref(b), _ref(ids), _ref(dst)), "mul_mat_id_silu_mul")
    return dst


def mul_mat_id_combine(as_down, b, ids, probs, resid=None, dst=None):
    """MUL_MAT_ID(down experts) over two slots + normalized top-2 weights + slot sum (+ residual) for one token in one launch"""
    dst = dst or Tensor(F32, [as_down.ne[1], 1])
    _l.check(_l.get().cllm_op_mul_mat_id_combine(None, _ref(as_down), _ref(b), _ref(ids), _ref(probs), _ref(resid), _ref(dst)), "mul_mat_id_combine")
    return dst


def moe_router(x, norm_w, eps, gate_w, k):
    """GenericSparseMLP's head for one token in one launch: (xnorm, probs, ids) = (RMS_NORM(x) * w, SOFT_MAX(gate . xnorm), TOP_K(probs, k))"""
    xnorm = Tensor(F32, [x.ne[0]]); probs = Tensor(F32, [gate_w.ne[1]]); ids = Tensor(I32, [k])
    _l.check(_l.get().cllm_op_moe_router(None, _ref(x), _ref(norm_w), float(eps), _ref(gate_w), _ref(xnorm), _ref(probs), _ref(ids)), "moe_router")
    return xnorm, probs, ids


def moe_router_gate_up(x, norm_w, eps, gate_w, as_gate, as_up, k):
    """router + the experts' gate / up / SiLU / MUL of one token in ONE launch: (probs, ids, g) -- the bits of moe_router followed by mul_mat_id_silu_mul"""
    L = _l.get()
    K, F, E = as_gate.ne[0], as_gate.ne[1], as_gate.ne[2]
    packed = Tensor(as_gate.type, [K, 2 * F, E])
    srcs = (C.c_void_p * 2)(as_gate.data_ptr().value, as_up.data_ptr().value)
    rows = (C.c_int64 * 2)(F * E, F * E)
    _l.check(L.cllm_pack_rows(None, packed.data_ptr(), srcs, rows, 2, as_gate.nb[1], 1), "pack_rows")
    probs = Tensor(F32, [E]); ids = Tensor(I32, [k]); dst = Tensor(F32, [F, k, 1])
    _l.check(L.cllm_op_moe_router_gate_up(None, _ref(x), _ref(norm_w), float(eps), _ref(gate_w), _ref(packed), _ref(probs), _ref(ids), _ref(dst)), "moe_router_gate_up")
    return probs, ids, dst


def moe_combine(experts, probs, ids, resid=None, dst=None):
    """GenericSparseMLP's tail: normalized top-k weights applied to the expert outputs, summed over the slots (+ residual)"""
    dst = dst or Tensor(F32, [experts.ne[0], experts.ne[2]])
    _l.check(_l.get().cllm_op_moe_combine(None, _ref(experts), _ref(probs), _ref(ids), _ref(resid), _ref(dst)), "moe_combine")
    return dst


def silu_mul(g, u, dst=None):
    dst = dst or Tensor(F32, g.ne)
    _l.check(_l.get().cllm_op_silu_mul(None, _ref(g), _ref(u), _ref(dst)), "silu_mul")
    return dst


def set_rows(dst, src, idx):
    """ggml::set_rows(ctx, a=dst, c=idx, b=src): writes into dst (a view of the KV cache)"""
    _l.check(_l.get().cllm_op_set_rows(None, _ref(src), _ref(idx), _ref(dst)), "set_rows")
    return dst


def cpy(src, dst):
    _l.check(_l.get().cllm_op_cpy(None, _ref(src), _ref(dst)), "cpy")
    return dst


def cont(a):
    return cpy(a, Tensor(a.type, a.ne))


def get_rows(a, idx, dst=None):
    dst = dst or Tensor(F32, [a.ne[0], idx.ne[0], idx.ne[1], idx.ne[2]])
    _l.check(_l.get().cllm_op_get_rows(None, _ref(a), _ref(idx), _ref(dst)), "get_rows")
    return dst


def sync():
    _l.check(_l.get().cllm_stream_sync(None), "sync")
