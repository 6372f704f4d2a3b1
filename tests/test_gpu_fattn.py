"""GPU parity of GGML_OP_FLASH_ATTN_EXT (cllm_op_flash_attn_ext) and of the fused prefill attention (cllm_op_attn_prefill).

TOLERANCE tier: the reference op has three summation orders of its own (one_chunk / tiled / split-KV, chosen by shape and thread
count; tests/test_oracle_vs_reference.py pins the oracle to the first bit-exactly and measures 4e-3 .. 2e-2 of max|out| between
them).  Two checks per case, both relative to max|out|:
  * against the oracle (oracle/ggml_oracle.c orc_flash_attn_ext, the reference's one_chunk order with its fp16 V accumulator): FA_ORACLE
  * against the same formula in float64 over the SAME converted operands (Q rounded to fp16 / quantize_row_q8_0 like the CPU,
    K / V dequantized): FA_EXACT -- what remains is the fp16 rounding of P before P.V and fp32 accumulation.
"""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from conftest import prefill_mode
from synth_helpers import rand_blocks

pytestmark = pytest.mark.gpu
FA_ORACLE = 2e-2
FA_EXACT = 2e-3


def rel_err(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) / (np.max(np.abs(b)) + 1e-30))


def make_case(kv_t, D, N, H, Hkv, n_kv, n_past, mask_mode, seed):
    r = np.random.default_rng(seed)
    q = (r.standard_normal((H, N, D)) * 1.5).astype(np.float32)
    if kv_t == O.F16:
        k = (r.standard_normal((Hkv, n_kv, D)) * 0.8).astype(np.float16)
        v = r.standard_normal((Hkv, n_kv, D)).astype(np.float16)
    else:
        k = rand_blocks(O.Q8_0, Hkv * n_kv, D, r).reshape(Hkv, n_kv, -1)
        v = rand_blocks(O.Q8_0, Hkv * n_kv, D, r).reshape(Hkv, n_kv, -1)
    mask = None
    if mask_mode == "causal":                     # CoreAttention::before_eval (src/layers.cpp:2585-2618)
        m = np.zeros((N, n_kv), np.float32)
        for j in range(N):
            m[j, 1 + j + n_past:] = -np.inf
        mask = m.astype(np.float16)
    elif mask_mode == "bias":                     # finite additive values and a fully masked leading stretch
        m = (r.standard_normal((N, n_kv)) * 0.5).astype(np.float32)
        m[:, : n_kv // 3] = -np.inf
        if N > 1:
            m[N // 2, :] = -np.inf                # a query with nothing visible: the CPU writes zeros (S == 0)
        mask = m.astype(np.float16)
    return q, k, v, mask


def deq(kv_t, a, D):
    if kv_t == O.F16:
        return a.astype(np.float64)
    flat = a.reshape(-1, a.shape[-1])
    return np.stack([O.dequantize(O.Q8_0, row, D) for row in flat]).reshape(a.shape[0], a.shape[1], D).astype(np.float64)


def exact(kv_t, q, k, v, mask, scale, D):
    H, N, _ = q.shape
    Hkv = k.shape[0]
    if kv_t == O.F16:
        qc = q.astype(np.float16).astype(np.float64)
    else:
        qc = np.stack([O.dequantize(O.Q8_0, O.quantize_q8_0(row), D) for row in q.reshape(-1, D)]).reshape(q.shape).astype(np.float64)
    K, V = deq(kv_t, k, D), deq(kv_t, v, D)
    out = np.zeros((N, H, D))
    for h in range(H):
        s = qc[h] @ K[h // (H // Hkv)].T * scale
        if mask is not None:
            s = s + mask.astype(np.float64)
        mx = np.max(s, axis=1, keepdims=True)
        mx[~np.isfinite(mx)] = 0.0
        p = np.exp(s - mx)
        l = p.sum(axis=1, keepdims=True)
        out[:, h, :] = np.where(l > 0, (p @ V[h // (H // Hkv)]) / np.where(l > 0, l, 1.0), 0.0)
    return out


def oracle_fa(kv_t, D, N, H, Hkv, n_kv, q, k, v, mask, scale):
    out = np.zeros((N, H, D), np.float32)
    rb = O.row_size(kv_t, D)
    e = 2 if kv_t == O.F16 else 34
    O.flash_attn_ext(O.tensor(q, O.F32, [D, N, H]), O.tensor(k, kv_t, [D, n_kv, Hkv], nb=[e, rb, rb * n_kv, rb * n_kv * Hkv]),
                     O.tensor(v, kv_t, [D, n_kv, Hkv], nb=[e, rb, rb * n_kv, rb * n_kv * Hkv]),
                     O.tensor(mask, O.F16, [n_kv, N]) if mask is not None else None, O.tensor(out, O.F32, [D, H, N]), scale)
    return out


def gpu_fa(gpu, kv_t, D, N, H, Hkv, n_kv, q, k, v, mask, scale):
    T = gpu.Tensor
    dq = T.from_numpy(q, gpu.F32, [D, N, H])
    dk = T.from_numpy(k if kv_t == O.F16 else k.reshape(-1), kv_t, [D, n_kv, Hkv])
    dv = T.from_numpy(v if kv_t == O.F16 else v.reshape(-1), kv_t, [D, n_kv, Hkv])
    dm = T.from_numpy(mask, gpu.F16, [n_kv, N]) if mask is not None else None
    return gpu.ops.flash_attention(dq, dk, dv, dm, scale).numpy().reshape(N, H, D)


CASES = [
    # kv type, D, N, H, Hkv, n_kv, n_past, mask
    (O.F16, 128, 1, 8, 2, 1, 0, "causal"),            # first token
    (O.F16, 128, 1, 8, 2, 38, 37, "causal"),          # decode, one tile
    (O.F16, 128, 1, 32, 8, 300, 299, "causal"),       # decode, split over 5 workgroups per kv head (llama3 head layout)
    (O.F16, 128, 1, 8, 8, 5000, 4999, None),          # decode, long: 64 splits of 2 tiles, no mask tensor, no GQA
    (O.F16, 64, 1, 12, 12, 77, 76, "causal"),         # head size 64 (gpt2-small-like), odd n_kv: unaligned mask rows
    (O.F16, 128, 3, 8, 2, 131, 128, "causal"),        # a few queries: still the packed (GQA rows) form
    (O.F16, 128, 16, 8, 2, 16, 0, "causal"),          # short prefill
    (O.F16, 128, 200, 8, 2, 200, 0, "causal"),        # prefill, ragged: 2 query blocks, 4 tiles
    (O.F16, 128, 130, 4, 4, 391, 261, "causal"),      # prefill continuing a cached context, odd sizes
    (O.F16, 64, 257, 4, 2, 257, 0, "causal"),
    (O.F16, 128, 40, 4, 2, 96, 56, "bias"),           # finite additive mask, fully masked tiles and one fully masked query
    (O.F16, 128, 1, 8, 2, 200, 199, "bias"),
    (O.Q8_0, 128, 1, 8, 2, 300, 299, "causal"),       # --cache_dtype q8_0
    (O.Q8_0, 128, 96, 8, 2, 160, 64, "causal"),
    (O.Q8_0, 64, 1, 4, 4, 1000, 999, None),
]


@pytest.mark.parametrize("kv_t,D,N,H,Hkv,n_kv,n_past,mask_mode", CASES)
def test_flash_attn_ext(gpu, kv_t, D, N, H, Hkv, n_kv, n_past, mask_mode):
    q, k, v, mask = make_case(kv_t, D, N, H, Hkv, n_kv, n_past, mask_mode, seed=n_kv * 7 + N)
    scale = 1.0 / np.sqrt(D)
    got = gpu_fa(gpu, kv_t, D, N, H, Hkv, n_kv, q, k, v, mask, scale)
    assert np.all(np.isfinite(got))
    want = oracle_fa(kv_t, D, N, H, Hkv, n_kv, q, k, v, mask, scale)
    ex = exact(kv_t, q, k, v, mask, scale, D)
    e_or, e_ex = rel_err(got, want), rel_err(got, ex)
    assert e_ex < FA_EXACT, (e_or, e_ex)
    assert e_or < FA_ORACLE, (e_or, e_ex)
    if mask_mode == "bias" and N > 1:
        assert np.all(got[N // 2] == 0.0)            # nothing visible -> zeros, like the CPU (S_inv = 0)


def test_flash_attn_ext_strided_cache_views(gpu):
    """K / V as chatllm passes them: views of the [k_hidden, max_len] caches (src/layers.cpp:3125-3160), q a permuted view"""
    T = gpu.Tensor
    D, N, H, Hkv, ML, n_past = 128, 5, 8, 2, 64, 20
    n_kv = n_past + N
    r = np.random.default_rng(5)
    q = r.standard_normal((N, H, D)).astype(np.float32)                       # [D, H, N] as the projection leaves it
    kc = (r.standard_normal((ML, Hkv * D)) * 0.8).astype(np.float16)          # K cache rows = positions
    vc = r.standard_normal((Hkv, ML, D)).astype(np.float16)                   # V cache [head_size, max_len, kv_heads]
    mask = np.zeros((N, n_kv), np.float32)
    for j in range(N):
        mask[j, 1 + j + n_past:] = -np.inf
    dq = T.from_numpy(q, gpu.F32, [D, H, N]).permute(0, 2, 1, 3)
    dk = T.from_numpy(kc, gpu.F16, [Hkv * D, ML]).view([D, n_kv, Hkv], [2, Hkv * D * 2, D * 2])
    dv = T.from_numpy(vc, gpu.F16, [D, ML, Hkv]).view([D, n_kv, Hkv], [2, D * 2, D * ML * 2])
    dm = T.from_numpy(mask.astype(np.float16), gpu.F16, [n_kv, N])
    got = gpu.ops.flash_attention(dq, dk, dv, dm, 1.0 / np.sqrt(D)).numpy().reshape(N, H, D)
    k = np.ascontiguousarray(kc[:n_kv].reshape(n_kv, Hkv, D).transpose(1, 0, 2))
    v = np.ascontiguousarray(vc[:, :n_kv])
    ex = exact(O.F16, np.ascontiguousarray(q.transpose(1, 0, 2)), k, v, mask.astype(np.float16), 1.0 / np.sqrt(D), D)
    assert rel_err(got, ex) < FA_EXACT


def test_flash_attn_ext_declines_what_it_does_not_take(gpu):
    T = gpu.Tensor
    q = T.from_numpy(np.zeros((2, 1, 80), np.float32), gpu.F32, [80, 1, 2])
    k = T.from_numpy(np.zeros((2, 4, 80), np.float16), gpu.F16, [80, 4, 2])
    with pytest.raises(Exception):
        gpu.ops.flash_attention(q, k, k, None, 1.0)                           # head size 80
    q = T.from_numpy(np.zeros((2, 1, 64), np.float32), gpu.F32, [64, 1, 2])
    k = T.from_numpy(np.zeros((2, 4, 64), np.float16), gpu.F16, [64, 4, 2])
    with pytest.raises(Exception):
        gpu.ops.flash_attention(q, k, k, None, 1.0, max_bias=8.0)             # ALiBi


@pytest.mark.parametrize("D,N,H,Hkv,n_past,ML", [(128, 64, 8, 2, 0, 64), (128, 200, 8, 2, 0, 256), (128, 130, 4, 4, 261, 400), (64, 300, 4, 2, 33, 336),
                                                 # cached lengths that are multiples of 8 from 256 on: K.Q by columns and V.P straight from global memory into the MFMA layout
                                                 # (k_mmf_exact_kq / _vp): ragged column tiles, a past, head size 64, GQA, a masked last step
                                                 (128, 320, 8, 2, 0, 320), (128, 200, 4, 4, 184, 384), (64, 296, 4, 2, 40, 336), (128, 1000, 2, 1, 24, 1024),
                                                 # 16 and 24 heads: the head index of a causal launch is permuted across the grid (an XCD keeps one K/V head) / left alone
                                                 (64, 264, 16, 4, 0, 264), (64, 136, 24, 8, 128, 264)])
def test_attn_prefill_against_the_node_sequence(gpu, D, N, H, Hkv, n_past, ML):
    """the fused prefill attention against the oracle's MUL_MAT + SCALE + DIAG_MASK_INF + SOFT_MAX + MUL_MAT over the cache views"""
    T = gpu.Tensor
    n_kv, KD = n_past + N, D * Hkv
    r = np.random.default_rng(N)
    q = (r.standard_normal((N, H, D)) * 1.5).astype(np.float32)
    kc = (r.standard_normal((ML, KD)) * 0.8).astype(np.float16)
    vc = r.standard_normal((KD, ML)).astype(np.float16)
    vc[:, n_kv:] = np.float16(np.nan)                                         # beyond the cached length: never read as a value
    scale = 1.0 / np.sqrt(D)
    sc = np.zeros((H, N, n_kv), np.float32)
    ctx = np.zeros((H, N, D), np.float32)
    Kv = O.tensor(kc, O.F16, [D, n_kv, Hkv], nb=[2, KD * 2, D * 2, KD * ML * 2])
    Qv = O.tensor(q, O.F32, [D, N, H], nb=[4, H * D * 4, D * 4, H * D * N * 4])
    S = O.tensor(sc, O.F32, [n_kv, N, H])
    O.mul_mat(Kv, Qv, S)
    O.scale(S, S, scale)
    O.diag_mask_inf(S, S, n_past)
    O.soft_max(S, None, S)
    Vv = O.tensor(vc, O.F16, [n_kv, D, Hkv], nb=[2, ML * 2, ML * D * 2, ML * KD * 2])
    O.mul_mat(Vv, S, O.tensor(ctx, O.F32, [D, N, H]))

    dq = T.from_numpy(q, gpu.F32, [D, H, N]).permute(0, 2, 1, 3)
    dk = T.from_numpy(kc, gpu.F16, [KD, ML]).view([D, n_kv, Hkv], [2, KD * 2, D * 2])
    dv = T.from_numpy(vc, gpu.F16, [ML, KD]).view([n_kv, D, Hkv], [2, ML * 2, ML * D * 2])
    got = gpu.ops.attn_prefill(dq, dk, dv, scale, n_past).numpy().reshape(H, N, D)
    # the default (prefill mode 1): K.Q -> soft_max -> V.P in the reference's order (mmf_exact.hip): every word equals the node sequence's
    assert np.array_equal(got.view(np.uint32), ctx.view(np.uint32)), rel_err(got, ctx)
    with prefill_mode(gpu, 0):             # CLLM_PREFILL=fast: ONE flash kernel, tolerance tier
        got = gpu.ops.attn_prefill(dq, dk, dv, scale, n_past).numpy().reshape(H, N, D)
    assert np.all(np.isfinite(got))
    assert rel_err(got, ctx) < FA_EXACT
