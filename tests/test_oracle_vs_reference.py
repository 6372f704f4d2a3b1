"""Pins oracle/ggml_oracle.c (our C restatement) against the REAL reference CPU backend
(oracle/_ref/libggml-cpu.so built from /root/reference by oracle/Makefile, driven through
oracle/ref_ops.c).  The reference ships no tests, so this is the parity pin.
Skipped where oracle/_ref has not been built (it always is in the build container and it
travels to the GPU box as a prebuilt .so)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
from synth_helpers import rand_blocks

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built")

P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
rng = np.random.default_rng(7)


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


@pytest.mark.parametrize("K", [32, 256, 4096, 14336])
def test_quantize_q8_0_bit_exact(K):
    R = O.ref()
    for scale in (1.0, 1e-3, 300.0):
        x = (rng.standard_normal(K) * scale).astype(np.float32)
        if K >= 256:
            x[:32] = 0.0                     # all-zero block
            x[40] = x[41] = -x[42]           # ties in |x|
            x[64:96] = np.arange(32, dtype=np.float32) + 0.5   # amax = 31.5 ... exact halves after scaling show the rounding mode
            x[96:128] = (np.arange(32, dtype=np.float32) - 16) * 0.5; x[96] = 127.0   # id == 1: every k+0.5 is a tie
        got = O.quantize_q8_0(x)
        ref = np.zeros_like(got)
        assert R.ref_quantize_cpu(O.Q8_0, P(x), P(ref), C.c_int64(K)) == 0
        assert np.array_equal(got, ref)
        got = O.quantize_q8_0(x, ref=True)
        assert R.ref_quantize_ref(O.Q8_0, P(x), P(ref), C.c_int64(K)) == 0
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("K", [32, 256, 4096, 14336])
def test_quantize_q8_1_bit_exact(K):
    """the activation format of Q4_1 weights: Q8_0's quants plus s = fp16(d * sum q)"""
    R = O.ref()
    for scale in (1.0, 1e-3, 300.0, 6e4):
        x = (rng.standard_normal(K) * scale).astype(np.float32)
        if K >= 256:
            x[:32] = 0.0
            x[64:96] = np.arange(32, dtype=np.float32) + 0.5
            x[96:128] = np.abs(x[96:128])             # all-positive block: the largest |s|
        got = O.quantize_q8_1(x)
        ref = np.zeros_like(got)
        assert R.ref_quantize_cpu(O.Q8_1, P(x), P(ref), C.c_int64(K)) == 0
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("K", [256, 4096, 14336])
def test_quantize_q8_K_bit_exact(K):
    R = O.ref()
    for scale in (1.0, 1e-4, 50.0):
        x = (rng.standard_normal(K) * scale).astype(np.float32)
        x[10] = -x[3]                    # same magnitude, opposite sign: the FIRST one decides the sign of iscale
        x[256 * (K // 256 - 1):] *= -1
        got = O.quantize_q8_K(x)
        ref = np.zeros_like(got)
        assert R.ref_quantize_cpu(O.Q8_K, P(x), P(ref), C.c_int64(K)) == 0
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("t", [O.Q8_0, O.Q4_0, O.Q4_1, O.Q5_0, O.Q5_1, O.Q4_K, O.F16])
def test_weight_quantizers_bit_exact(t):
    """from_float_ref of the weight types (the loader's re-quantization, src/chat.cpp:1246-1279): every byte equals the reference's"""
    R = O.ref()
    for K in (256, 4096, 14336):
        for scale in (1.0, 1e-3, 40.0):
            x = (rng.standard_normal(K) * scale).astype(np.float32)
            x[:32] = 0.0                                   # an all-zero block / sub-block
            x[32:64] = 0.37                                # a constant one (max == min)
            x[64:96] = np.abs(x[64:96]) + 0.1              # all positive: the minimum is clamped to 0
            x[100] = -x[101]                               # a tie in |x|: the first one decides
            if K > 256:
                x[256:512] = np.round(x[256:512] * 4) / 4  # values on a grid: exact halves in the rounding
            got = O.quantize_ref(t, x)
            ref = np.zeros_like(got)
            assert R.ref_quantize_ref(t, P(x), P(ref), C.c_int64(K)) == 0
            assert np.array_equal(got, ref), (K, scale, int(np.argmax(got != ref)))


@pytest.mark.parametrize("t", [O.Q4_0, O.Q4_1, O.Q8_0, O.Q4_K, O.Q5_K, O.Q6_K, O.Q5_0, O.Q5_1, O.Q2_K, O.Q3_K, O.IQ4_NL, O.MXFP4, O.IQ4_XS, O.TQ1_0, O.TQ2_0, O.IQ2_XXS, O.IQ2_XS, O.IQ2_S, O.IQ3_XXS, O.IQ3_S, O.IQ1_S, O.IQ1_M])
def test_dequantize_bit_exact(t):
    R = O.ref()
    K = 2048
    w = rand_blocks(t, 1, K, rng)
    got = O.dequantize(t, w, K)
    ref = np.zeros(K, np.float32)
    assert R.ref_dequantize(t, P(w), P(ref), C.c_int64(K)) == 0
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("t", [O.Q4_0, O.Q4_1, O.Q8_0, O.Q4_K])
def test_vec_dot_matches_reference(t):
    R = O.ref()
    K = 4096
    w = rand_blocks(t, 1, K, rng)
    x = rng.standard_normal(K).astype(np.float32)
    a = O.quantize_q8_K(x) if t == O.Q4_K else O.quantize_q8_1(x) if t == O.Q4_1 else O.quantize_q8_0(x)
    got, isums = O.vec_dot(t, K, w, a)
    s = C.c_float()
    assert R.ref_vec_dot(t, C.c_int64(K), P(w), P(a), C.byref(s)) == 0
    # the generic-branch restatement: identical integer sums, different fp32 summation order than the AVX2 branch the reference build takes
    assert abs(got - s.value) <= 2e-5 * (abs(s.value) + 1.0)
    # the AVX2-order restatement (what mul_mat and the whole-model walk use): the same bits
    fn = {O.Q4_0: "orc_vec_dot_q4_0_q8_0_avx2", O.Q8_0: "orc_vec_dot_q8_0_q8_0_avx2", O.Q4_1: "orc_vec_dot_q4_1_q8_1_avx2", O.Q4_K: "orc_vec_dot_q4_K_q8_K_avx2"}[t]
    f = getattr(O.lib(), fn); f.restype = C.c_float; f.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
    for _ in range(50):
        w = rand_blocks(t, 1, K, rng)
        x = (rng.standard_normal(K) * rng.choice([1e-2, 1.0, 30.0])).astype(np.float32)
        a = O.quantize_q8_K(x) if t == O.Q4_K else O.quantize_q8_1(x) if t == O.Q4_1 else O.quantize_q8_0(x)
        assert R.ref_vec_dot(t, C.c_int64(K), P(w), P(a), C.byref(s)) == 0
        assert np.float32(f(C.c_int64(K), P(np.ascontiguousarray(w)), P(np.ascontiguousarray(a)))).view(np.uint32) == np.float32(s.value).view(np.uint32)
    # integer sums against a direct dequantized-integer computation
    assert isums.dtype == np.int32 and np.all(np.abs(isums) < 2**31 - 1)


@pytest.mark.parametrize("t", [O.Q5_K, O.Q6_K])
def test_vec_dot_k_quants_avx2_order_bit_exact(t):
    """Q5_K / Q6_K: only the AVX2 order is restated; 8 lane accumulators + (Q5_K) the scalar mins chain"""
    R = O.ref()
    fn = {O.Q5_K: "orc_vec_dot_q5_K_q8_K_avx2", O.Q6_K: "orc_vec_dot_q6_K_q8_K_avx2"}[t]
    f = getattr(O.lib(), fn); f.restype = C.c_float; f.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
    s = C.c_float()
    for K in (256, 4096, 14336):
        for _ in range(30):
            w = rand_blocks(t, 1, K, rng)
            x = (rng.standard_normal(K) * rng.choice([1e-2, 1.0, 30.0])).astype(np.float32)
            a = O.quantize_q8_K(x)
            assert R.ref_vec_dot(t, C.c_int64(K), P(w), P(a), C.byref(s)) == 0
            assert np.float32(f(C.c_int64(K), P(np.ascontiguousarray(w)), P(np.ascontiguousarray(a)))).view(np.uint32) == np.float32(s.value).view(np.uint32)


@pytest.mark.parametrize("t", [O.Q5_0, O.Q5_1, O.IQ4_NL, O.MXFP4, O.Q2_K, O.Q3_K, O.IQ4_XS, O.TQ1_0, O.TQ2_0, O.IQ2_XXS, O.IQ2_XS, O.IQ2_S, O.IQ3_XXS, O.IQ3_S, O.IQ1_S, O.IQ1_M])
def test_vec_dot_other_formats_avx2_order_bit_exact(t):
    """Q5_0 / Q5_1 / IQ4_NL / MXFP4 / Q2_K / Q3_K: the x86 AVX2 order restated (8 lane accumulators; the codebook formats pair their blocks over two of them and
    finish an unpaired block in scalar code)"""
    R = O.ref()
    fn = {O.Q5_0: "orc_vec_dot_q5_0_q8_0_avx2", O.Q5_1: "orc_vec_dot_q5_1_q8_1_avx2", O.IQ4_NL: "orc_vec_dot_iq4_nl_q8_0_avx2", O.MXFP4: "orc_vec_dot_mxfp4_q8_0_avx2",
          O.Q2_K: "orc_vec_dot_q2_K_q8_K_avx2", O.Q3_K: "orc_vec_dot_q3_K_q8_K_avx2", O.IQ4_XS: "orc_vec_dot_iq4_xs_q8_K_avx2",
          O.TQ1_0: "orc_vec_dot_tq1_0_q8_K_avx2", O.TQ2_0: "orc_vec_dot_tq2_0_q8_K_avx2",
          O.IQ1_S: "orc_vec_dot_iq1_s_q8_K_avx2", O.IQ1_M: "orc_vec_dot_iq1_m_q8_K_avx2"}.get(t, "orc_vec_dot_iq_grid_q8_K_avx2")
    grid = t in (O.IQ2_XXS, O.IQ2_XS, O.IQ2_S, O.IQ3_XXS, O.IQ3_S)
    f = getattr(O.lib(), fn); f.restype = C.c_float
    f.argtypes = ([C.c_int] if grid else []) + [C.c_int64, C.c_void_p, C.c_void_p] + ([C.c_int] if t == O.IQ4_NL else [])
    s = C.c_float()
    for K in ((256, 4096, 14336) if O.BLCK[t] == 256 else (32, 96, 256, 4096, 14336, 14368)):      # (odd block counts: the scalar tail of the codebook formats)
        for _ in range(30):
            w = rand_blocks(t, 1, K, rng)
            x = (rng.standard_normal(K) * rng.choice([1e-2, 1.0, 30.0])).astype(np.float32)
            a = O.quantize_q8_K(x) if O.BLCK[t] == 256 else O.quantize_q8_1(x) if t == O.Q5_1 else O.quantize_q8_0(x)
            assert R.ref_vec_dot(t, C.c_int64(K), P(w), P(a), C.byref(s)) == 0
            args = ([C.c_int(t)] if grid else []) + [C.c_int64(K), P(np.ascontiguousarray(w)), P(np.ascontiguousarray(a))] + ([C.c_int(0)] if t == O.IQ4_NL else [])
            assert np.float32(f(*args)).view(np.uint32) == np.float32(s.value).view(np.uint32), (K, f(*args), s.value)


@pytest.mark.parametrize("t,K,N,M", [(O.Q5_0, 256, 40, 1), (O.Q5_0, 512, 17, 7), (O.Q5_0, 4096, 64, 33), (O.Q5_1, 256, 40, 1), (O.Q5_1, 1024, 19, 6),
                                     (O.IQ4_NL, 256, 40, 1), (O.IQ4_NL, 512, 17, 7), (O.IQ4_NL, 96, 24, 1), (O.IQ4_NL, 96, 24, 5), (O.IQ4_NL, 4096, 64, 33),
                                     (O.MXFP4, 256, 40, 1), (O.MXFP4, 96, 17, 7), (O.IQ4_XS, 512, 48, 1), (O.IQ4_XS, 1024, 33, 5), (O.IQ4_XS, 256, 16, 40),
                                     (O.TQ1_0, 512, 48, 1), (O.TQ1_0, 1024, 33, 5), (O.TQ1_0, 256, 16, 40), (O.TQ2_0, 512, 48, 1), (O.TQ2_0, 1024, 33, 5), (O.TQ2_0, 256, 16, 40),
                                     (O.IQ2_XXS, 512, 48, 1), (O.IQ2_XXS, 1024, 33, 5), (O.IQ2_XS, 512, 48, 1), (O.IQ2_XS, 256, 16, 40), (O.IQ2_S, 512, 48, 1), (O.IQ2_S, 1024, 33, 5),
                                     (O.IQ3_XXS, 512, 48, 1), (O.IQ3_XXS, 256, 16, 40), (O.IQ3_S, 512, 48, 1), (O.IQ3_S, 1024, 33, 5),
                                     (O.IQ1_S, 512, 48, 1), (O.IQ1_S, 1024, 33, 5), (O.IQ1_S, 256, 16, 40), (O.IQ1_M, 512, 48, 1), (O.IQ1_M, 1024, 33, 5), (O.IQ1_M, 256, 16, 40), (O.Q2_K, 512, 48, 1), (O.Q2_K, 1024, 33, 5), (O.Q3_K, 512, 48, 1), (O.Q3_K, 2048, 17, 7),
                                     (O.Q5_K, 512, 48, 1), (O.Q5_K, 1024, 33, 5), (O.Q6_K, 512, 48, 1), (O.Q6_K, 2048, 17, 7), (O.Q6_K, 256, 16, 40),
                                     (O.Q4_K, 512, 48, 1), (O.Q4_K, 1024, 33, 5), (O.Q4_K, 4096, 16, 40), (O.Q4_0, 256, 40, 1), (O.Q4_0, 512, 17, 7), (O.Q4_0, 4096, 64, 33),
                                     (O.Q4_1, 256, 40, 1), (O.Q4_1, 1024, 19, 6),
                                     (O.Q8_0, 256, 40, 1), (O.Q8_0, 1024, 31, 3), (O.Q8_0, 512, 40, 19), (O.F16, 128, 50, 3), (O.F32, 96, 20, 2),
                                     (O.F16, 128, 500, 1), (O.F16, 128, 64, 16), (O.F16, 128, 63, 16), (O.F16, 1000, 128, 4), (O.F16, 1001, 128, 5), (O.F16, 77, 128, 1),
                                     (O.F32, 4096, 8, 3), (O.F32, 4096, 6, 3), (O.F32, 100, 7, 1)])
def test_mul_mat(t, K, N, M):
    R = O.ref()
    if t in (O.F16, O.F32):
        w = rng.standard_normal((N, K)).astype(O.NP_OF[t])
    else:
        w = rand_blocks(t, N, K, rng)
    x = rng.standard_normal((M, K)).astype(np.float32)
    ref = np.zeros((M, N), np.float32)
    assert R.ref_mul_mat(t, C.c_int64(K), C.c_int64(N), C.c_int64(M), C.c_int64(1), C.c_int64(1), P(w), P(x), P(ref)) == 0
    got = np.zeros((M, N), np.float32)
    O.mul_mat(O.tensor(w, t, [K, N]), O.tensor(x, O.F32, [K, M]), O.tensor(got, O.F32, [N, M]))
    # every path of the reference's mul_mat is restated in its own order: the vec_dot loop (AVX2 branches), tinyBLAS_Q0_AVX for Q4_0 / Q8_0
    # with M >= 2 (the same per-element chain), tinyBLAS<8> for F16 / F32 with M >= 2, K % 8 == 0, N % 4 == 0 -- BIT-IDENTICAL
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), rel_err(got, ref)


def test_mul_mat_broadcast_heads():
    """GQA broadcast: src0 [K, N, 2] against src1 [K, M, 6]"""
    R = O.ref()
    K, N, M = 64, 24, 3
    w = rng.standard_normal((2, N, K)).astype(np.float16)
    x = rng.standard_normal((6, M, K)).astype(np.float32)
    ref = np.zeros((6, M, N), np.float32)
    assert R.ref_mul_mat(O.F16, C.c_int64(K), C.c_int64(N), C.c_int64(M), C.c_int64(2), C.c_int64(6), P(w), P(x), P(ref)) == 0
    got = np.zeros_like(ref)
    O.mul_mat(O.tensor(w, O.F16, [K, N, 2]), O.tensor(x, O.F32, [K, M, 6]), O.tensor(got, O.F32, [N, M, 6]))
    assert rel_err(got, ref) < 1e-5


@pytest.mark.parametrize("t", [O.Q4_K, O.Q8_0, O.Q4_1, O.Q4_0, O.Q5_K, O.Q6_K, O.Q2_K, O.Q3_K, O.Q5_0, O.Q5_1, O.IQ4_NL, O.MXFP4, O.IQ4_XS, O.TQ1_0, O.TQ2_0, O.IQ2_XXS, O.IQ2_XS, O.IQ2_S, O.IQ3_XXS, O.IQ3_S, O.IQ1_S, O.IQ1_M])
def test_mul_mat_id(t):
    R = O.ref()
    K, N, E, U, T = 512, 24, 4, 2, 3
    w = rand_blocks(t, N * E, K, rng)
    for nb1 in (1, U):
        x = rng.standard_normal((T, nb1, K)).astype(np.float32)
        ids = rng.integers(0, E, (T, U)).astype(np.int32)
        ref = np.zeros((T, U, N), np.float32)
        assert R.ref_mul_mat_id(t, C.c_int64(K), C.c_int64(N), C.c_int64(E), C.c_int64(nb1), C.c_int64(U), C.c_int64(T), P(w), P(x), P(ids), P(ref)) == 0
        got = np.zeros_like(ref)
        O.mul_mat_id(O.tensor(w, t, [K, N, E]), O.tensor(x, O.F32, [K, nb1, T]), O.tensor(ids, O.I32, [U, T]), O.tensor(got, O.F32, [N, U, T]))
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), rel_err(got, ref)        # vec_dot per (token, slot): the AVX2 order restated


@pytest.mark.parametrize("n0", [8, 100, 4096])
def test_rms_norm_bit_exact(n0):
    R = O.ref()
    x = rng.standard_normal((3, 5, n0)).astype(np.float32)
    ref = np.zeros_like(x)
    assert R.ref_unary(0, C.c_int64(n0), C.c_int64(5), C.c_int64(3), P(x), P(ref), C.c_float(1e-5), C.c_int(0)) == 0
    got = np.zeros_like(x)
    O.rms_norm(O.tensor(x, O.F32, [n0, 5, 3]), O.tensor(got, O.F32, [n0, 5, 3]), 1e-5)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("n0", [4096, 1000])
def test_rms_norm_on_a_rounding_boundary_is_the_references_serial_sum(n0):
    """rows whose mean of squares sits on a float rounding boundary (synth_helpers.rms_boundary_rows): the restatement adds in the reference's serial order
    (ops.cpp:3736-3741), so it agrees with libggml-cpu.so where a pairwise sum would not -- these rows are what the GPU's order-exact RMS_NORM is tested on"""
    from synth_helpers import rms_boundary_rows
    R = O.ref()
    rows = 16
    x = rms_boundary_rows(n0, rows, np.random.default_rng(n0 + 1))
    ref = np.zeros_like(x)
    assert R.ref_unary(0, C.c_int64(n0), C.c_int64(rows), C.c_int64(1), P(x), P(ref), C.c_float(1e-5), C.c_int(0)) == 0
    got = np.zeros_like(x)
    O.rms_norm(O.tensor(x, O.F32, [n0, rows]), O.tensor(got, O.F32, [n0, rows]), 1e-5)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    pairwise_wrong = 0
    for r in x:
        sq = (r * r).astype(np.float64)
        ser = 0.0
        for v in sq:
            ser += v
        pairwise_wrong += np.float32(ser / n0) != np.float32(float(np.sum(sq)) / n0)      # (float mean of the serial sum vs of numpy's pairwise sum)
    assert pairwise_wrong >= 3, pairwise_wrong          # the order decides on these rows


@pytest.mark.parametrize("n0", [7, 8, 61, 256])
def test_silu_bit_exact(n0):
    R = O.ref()
    x = (rng.standard_normal((4, n0)) * 4).astype(np.float32)
    x[0, 0] = 100.0
    x[1, 0] = -100.0
    ref = np.zeros_like(x)
    assert R.ref_unary(1, C.c_int64(n0), C.c_int64(4), C.c_int64(1), P(x), P(ref), C.c_float(0), C.c_int(0)) == 0
    got = np.zeros_like(x)
    O.silu(O.tensor(x, O.F32, [n0, 4]), O.tensor(got, O.F32, [n0, 4]))
    nv = n0 & ~7                       # vector body is bit exact; the scalar tail uses libm expf on both sides
    assert np.array_equal(got[:, :nv].view(np.uint32), ref[:, :nv].view(np.uint32))
    assert np.allclose(got, ref, rtol=1e-6, atol=0)


@pytest.mark.parametrize("n0", [5, 8, 33, 1024])
def test_soft_max_bit_exact(n0):
    R = O.ref()
    x = (rng.standard_normal((2, 3, n0)) * 3).astype(np.float32)
    ref = np.zeros_like(x)
    assert R.ref_unary(2, C.c_int64(n0), C.c_int64(3), C.c_int64(2), P(x), P(ref), C.c_float(0), C.c_int(0)) == 0
    got = np.zeros_like(x)
    O.soft_max(O.tensor(x, O.F32, [n0, 3, 2]), None, O.tensor(got, O.F32, [n0, 3, 2]))
    assert np.allclose(got, ref, rtol=2e-7, atol=0)
    if n0 % 8 == 0:
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_soft_max_ext_mask():
    R = O.ref()
    n0, n1, n2 = 40, 6, 3
    x = rng.standard_normal((n2, n1, n0)).astype(np.float32)
    mask = np.where(rng.random((n1, n0)) < 0.3, -np.inf, 0.0).astype(np.float32)
    mask[:, 0] = 0.0
    for f16 in (0, 1):
        mk = mask.astype(np.float16) if f16 else mask
        ref = np.zeros_like(x)
        assert R.ref_soft_max_ext(C.c_int64(n0), C.c_int64(n1), C.c_int64(n2), P(x), P(mk), C.c_int(f16), C.c_float(0.125), P(ref)) == 0
        got = np.zeros_like(x)
        O.soft_max(O.tensor(x, O.F32, [n0, n1, n2]), O.tensor(mk, O.F16 if f16 else O.F32, [n0, n1]), O.tensor(got, O.F32, [n0, n1, n2]), scale=0.125)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_diag_mask_and_scale():
    R = O.ref()
    x = rng.standard_normal((2, 5, 9)).astype(np.float32)
    ref = np.zeros_like(x)
    got = np.zeros_like(x)
    assert R.ref_unary(3, C.c_int64(9), C.c_int64(5), C.c_int64(2), P(x), P(ref), C.c_float(0), C.c_int(4)) == 0
    O.diag_mask_inf(O.tensor(x, O.F32, [9, 5, 2]), O.tensor(got, O.F32, [9, 5, 2]), 4)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert R.ref_unary(4, C.c_int64(9), C.c_int64(5), C.c_int64(2), P(x), P(ref), C.c_float(0.088388), C.c_int(0)) == 0
    O.scale(O.tensor(x, O.F32, [9, 5, 2]), O.tensor(got, O.F32, [9, 5, 2]), 0.088388)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_add_mul_broadcast():
    R = O.ref()
    a = rng.standard_normal((3, 4, 16)).astype(np.float32)
    b = rng.standard_normal((1, 1, 16)).astype(np.float32)
    for op, fn in ((0, O.add), (1, O.mul)):
        ref = np.zeros_like(a)
        assert R.ref_binary(op, C.c_int64(16), C.c_int64(4), C.c_int64(3), P(a), C.c_int64(16), C.c_int64(1), C.c_int64(1), P(b), P(ref)) == 0
        got = np.zeros_like(a)
        fn(O.tensor(a, O.F32, [16, 4, 3]), O.tensor(b, O.F32, [16, 1, 1]), O.tensor(got, O.F32, [16, 4, 3]))
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("mode,hd,n_dims,ff", [(0, 128, 128, False), (2, 128, 128, False), (0, 64, 32, False), (2, 64, 64, True)])
def test_rope(mode, hd, n_dims, ff):
    R = O.ref()
    heads, qlen = 3, 5
    x = rng.standard_normal((qlen, heads, hd)).astype(np.float32)
    pos = np.array([0, 1, 7, 100, 4095], np.int32)
    ffv = (1.0 + rng.random(n_dims // 2)).astype(np.float32) if ff else None
    ref = np.zeros_like(x)
    assert R.ref_rope(C.c_int64(hd), C.c_int64(heads), C.c_int64(qlen), P(x), P(pos), P(ffv) if ff else None, C.c_int(n_dims), C.c_int(mode),
                      C.c_int(0), C.c_float(500000.0), C.c_float(1.0), C.c_float(0.0), C.c_float(1.0), C.c_float(0.0), C.c_float(0.0), P(ref)) == 0
    got = np.zeros_like(x)
    O.rope(O.tensor(x, O.F32, [hd, heads, qlen]), pos, ffv, O.tensor(got, O.F32, [hd, heads, qlen]), n_dims, mode, 500000.0)
    # same libm, the same iterated theta, and the rotation as the reference binary has it (gcc, -ffp-contract=fast: one fma around the rounded x1 product)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), int(np.sum(got.view(np.uint32) != ref.view(np.uint32)))


def test_rope_yarn():
    R = O.ref()
    hd, heads, qlen = 64, 2, 4
    x = rng.standard_normal((qlen, heads, hd)).astype(np.float32)
    pos = np.array([0, 3, 17, 50, 333, 1000, 4095, 20000], np.int32)
    qlen = len(pos)
    x = rng.standard_normal((qlen, heads, hd)).astype(np.float32)
    ref = np.zeros_like(x)
    assert R.ref_rope(C.c_int64(hd), C.c_int64(heads), C.c_int64(qlen), P(x), P(pos), None, C.c_int(hd), C.c_int(2), C.c_int(4096),
                      C.c_float(10000.0), C.c_float(0.25), C.c_float(1.0), C.c_float(1.2), C.c_float(32.0), C.c_float(1.0), P(ref)) == 0
    got = np.zeros_like(x)
    O.rope(O.tensor(x, O.F32, [hd, heads, qlen]), pos, None, O.tensor(got, O.F32, [hd, heads, qlen]), hd, 2, 10000.0,
           n_ctx_orig=4096, freq_scale=0.25, ext_factor=1.0, attn_factor=1.2, beta_fast=32.0, beta_slow=1.0)
    # the YaRN mixing and the magnitude correction as gcc contracted them (two fmas): bit for bit
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), int(np.sum(got.view(np.uint32) != ref.view(np.uint32)))


@pytest.mark.parametrize("dst_t,i64", [(O.F16, 0), (O.F16, 1), (O.F32, 0)])
def test_set_rows(dst_t, i64):
    R = O.ref()
    n0, rows, n = 48, 20, 6
    src = rng.standard_normal((n, n0)).astype(np.float32)
    idx = rng.permutation(rows)[:n].astype(np.int64 if i64 else np.int32)
    dst0 = rng.standard_normal((rows, n0)).astype(O.NP_OF[dst_t])
    ref = dst0.copy()
    assert R.ref_set_rows(dst_t, C.c_int64(n0), C.c_int64(rows), C.c_int64(n), P(src), P(idx), C.c_int(i64), P(ref)) == 0
    got = dst0.copy()
    O.set_rows(O.tensor(src, O.F32, [n0, n]), O.tensor(idx, O.I64 if i64 else O.I32, [n]), O.tensor(got, dst_t, [n0, rows]))
    assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))


def test_cpy_v_cache_transposed():
    R = O.ref()
    KD, qlen, ML, n_past = 32, 5, 24, 7
    v = rng.standard_normal((qlen, KD)).astype(np.float32)
    cache0 = rng.standard_normal((KD, ML)).astype(np.float16)
    ref = cache0.copy()
    assert R.ref_cpy_v_cache(C.c_int64(KD), C.c_int64(qlen), C.c_int64(ML), C.c_int64(n_past), P(v), P(ref)) == 0
    got = cache0.copy()
    src = O.tensor(v, O.F32, [qlen, KD], nb=[KD * 4, 4, KD * qlen * 4, KD * qlen * 4])          # transpose(v)
    dst = O.tensor(got, O.F16, [qlen, KD], nb=[2, ML * 2, ML * KD * 2, ML * KD * 2], offset=n_past * 2)
    O.cpy(src, dst)
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))


@pytest.mark.parametrize("t", [O.Q4_0, O.Q8_0, O.Q4_K, O.Q5_K, O.Q6_K, O.F16])
def test_get_rows(t):
    R = O.ref()
    n0, rows, n = 512, 30, 7
    table = rng.standard_normal((rows, n0)).astype(np.float16) if t == O.F16 else rand_blocks(t, rows, n0, rng)
    ids = rng.integers(0, rows, n).astype(np.int32)
    ref = np.zeros((n, n0), np.float32)
    assert R.ref_get_rows(t, C.c_int64(n0), C.c_int64(rows), P(table), C.c_int64(n), P(ids), P(ref)) == 0
    got = np.zeros_like(ref)
    O.get_rows(O.tensor(table, t, [n0, rows]), O.tensor(ids, O.I32, [n]), O.tensor(got, O.F32, [n0, n]))
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("qlen,n_past", [(1, 0), (1, 37), (6, 0), (5, 11)])
def test_attention_composite(qlen, n_past):
    """the exact node sequence chatllm emits for eager attention over strided cache views"""
    R = O.ref()
    hd, nh, nkv, ML = 64, 4, 2, 64
    KD, n_kv = hd * nkv, n_past + qlen
    q = rng.standard_normal((qlen, nh, hd)).astype(np.float32)
    kc = rng.standard_normal((ML, KD)).astype(np.float16)
    vc = rng.standard_normal((KD, ML)).astype(np.float16)
    ref = np.zeros((qlen, nh * hd), np.float32)
    sref = np.zeros((nh, qlen, n_kv), np.float32)
    assert R.ref_attention(C.c_int64(hd), C.c_int64(nh), C.c_int64(nkv), C.c_int64(qlen), C.c_int64(n_past), C.c_int64(ML), P(q), P(kc), P(vc), P(ref), P(sref)) == 0

    sc = np.zeros((nh, qlen, n_kv), np.float32)
    ctx = np.zeros((nh, qlen, hd), np.float32)
    Kv = O.tensor(kc, O.F16, [hd, n_kv, nkv], nb=[2, KD * 2, hd * 2, KD * ML * 2])
    Qv = O.tensor(q, O.F32, [hd, qlen, nh], nb=[4, nh * hd * 4, hd * 4, nh * hd * qlen * 4])
    S = O.tensor(sc, O.F32, [n_kv, qlen, nh])
    O.mul_mat(Kv, Qv, S)
    assert rel_err(sc, sref) < 1e-5
    O.scale(S, S, 1.0 / np.sqrt(hd))
    O.diag_mask_inf(S, S, n_past)
    O.soft_max(S, None, S)
    Vv = O.tensor(vc, O.F16, [n_kv, hd, nkv], nb=[2, ML * 2, ML * hd * 2, ML * KD * 2])
    O.mul_mat(Vv, S, O.tensor(ctx, O.F32, [hd, qlen, nh]))
    got = np.ascontiguousarray(ctx.transpose(1, 0, 2)).reshape(qlen, nh * hd)
    assert rel_err(got, ref) < 2e-5


def test_moe_router_ops_bit_exact():
    """SUM_ROWS (double accumulator), DIV (IEEE) and TOP_K (descending, first two swapped) as GenericSparseMLP::forward uses them"""
    R = O.ref()
    for n0, n1, n2, k in ((8, 5, 1, 2), (8, 1, 1, 2), (64, 7, 2, 6), (3, 4, 1, 1), (160, 3, 1, 8)):
        x = rng.standard_normal((n2, n1, n0)).astype(np.float32)
        x = np.exp(x) / np.exp(x).sum(-1, keepdims=True).astype(np.float32)          # router probabilities
        ref = np.zeros((n2, n1, 1), np.float32)
        assert R.ref_sum_rows(C.c_int64(n0), C.c_int64(n1), C.c_int64(n2), P(x), P(ref)) == 0
        got = np.zeros_like(ref)
        O.sum_rows(O.tensor(x, O.F32, [n0, n1, n2]), O.tensor(got, O.F32, [1, n1, n2]))
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        refk = np.zeros((n2, n1, k), np.int32)
        assert R.ref_top_k(C.c_int64(n0), C.c_int64(n1), C.c_int64(n2), P(x), C.c_int(k), P(refk)) == 0
        gotk = np.zeros_like(refk)
        O.top_k(O.tensor(x, O.F32, [n0, n1, n2]), O.tensor(gotk, O.I32, [k, n1, n2]))
        assert np.array_equal(gotk, refk)
        y = (np.abs(rng.standard_normal((n2, n1, 1))) + 0.1).astype(np.float32)
        refd = np.zeros_like(x)
        assert R.ref_binary(2, C.c_int64(n0), C.c_int64(n1), C.c_int64(n2), P(x), C.c_int64(1), C.c_int64(n1), C.c_int64(n2), P(y), P(refd)) == 0
        gotd = np.zeros_like(x)
        O.div(O.tensor(x, O.F32, [n0, n1, n2]), O.tensor(y, O.F32, [1, n1, n2]), O.tensor(gotd, O.F32, [n0, n1, n2]))
        assert np.array_equal(gotd.view(np.uint32), refd.view(np.uint32))


@pytest.mark.parametrize("arch,over,wt", [("qwen2", dict(qkv_bias=1, rope_mode=2, rope_theta=1e6), O.Q4_K), ("qwen2", dict(qkv_bias=1, rope_mode=2, rope_theta=1e6), O.Q8_0),
                                          ("llama3", {}, O.Q4_1)])
def test_oracle_whole_model_vs_reference_host_live(pkg, tmp_path, arch, over, wt):
    """the oracle's whole-model walk against the reference HOST run here (oracle/_ref/ref_chat on a synthetic GGMM file): the Qwen2
    architecture (q/k/v biases, NEOX RoPE -- BASELINE cfg4) next to the Llama-3 fixtures of tests/golden"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref_chat = os.path.join(root, "oracle", "_ref", "ref_chat")
    if not os.path.exists(ref_chat):
        pytest.skip("oracle/_ref/ref_chat not built")
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_ggmm
    cfg = pkg.synth.config("tiny", max_len=64, **over)
    mp, lp = str(tmp_path / "m.bin"), str(tmp_path / "l.bin")
    make_ggmm.write_model(mp, cfg, wt, seed=1234, arch=arch)
    prompt = [5, 9, 42, 300, 7, 99, 250, 12, 100]
    ids = subprocess.check_output([ref_chat, mp, "cpu", "4", "8", lp] + [str(p) for p in prompt], stderr=subprocess.DEVNULL, text=True).split()
    logits = np.fromfile(lp, np.float32).reshape(9, cfg["vocab"])
    m = O.Llama(cfg, pkg.synth.make_model(cfg, wt, seed=1234))
    lg = m.forward(np.array(prompt, np.int32))
    for s in range(9):          # bit-identical logits, hence identical greedy ids
        assert np.array_equal(lg.view(np.uint32), logits[s].view(np.uint32)), (s, float(np.max(np.abs(lg - logits[s]))))
        assert int(np.argmax(lg)) == int(ids[s])
        if s < 8:
            lg = m.forward([int(ids[s])])


def _fa_case(kv_t, D, N, H, Hkv, n_kv, n_past, masked, seed=3):
    r = np.random.default_rng(seed)
    q = r.standard_normal((H, N, D)).astype(np.float32)
    if kv_t == O.F16:
        k = (r.standard_normal((Hkv, n_kv, D)) * 0.7).astype(np.float16)
        v = r.standard_normal((Hkv, n_kv, D)).astype(np.float16)
    else:
        k = rand_blocks(O.Q8_0, Hkv * n_kv, D, r).reshape(Hkv, n_kv, -1)
        v = rand_blocks(O.Q8_0, Hkv * n_kv, D, r).reshape(Hkv, n_kv, -1)
    mask = None
    if masked:                                   # CoreAttention::before_eval (src/layers.cpp:2585-2618): causal, -inf above the diagonal
        m = np.zeros((N, n_kv), np.float32)
        for j in range(N):
            m[j, 1 + j + n_past:] = -np.inf
        mask = m.astype(np.float16)
    return q, k, v, mask


def _fa_oracle(kv_t, D, N, H, Hkv, n_kv, q, k, v, mask, scale):
    out = np.zeros((N, H, D), np.float32)
    rb = O.row_size(kv_t, D)
    O.flash_attn_ext(O.tensor(q, O.F32, [D, N, H]), O.tensor(k, kv_t, [D, n_kv, Hkv], nb=[O.row_size(kv_t, 1) if kv_t == O.F16 else 34, rb, rb * n_kv, rb * n_kv * Hkv]),
                     O.tensor(v, kv_t, [D, n_kv, Hkv], nb=[O.row_size(kv_t, 1) if kv_t == O.F16 else 34, rb, rb * n_kv, rb * n_kv * Hkv]),
                     O.tensor(mask, O.F16, [n_kv, N]) if mask is not None else None, O.tensor(out, O.F32, [D, H, N]), scale)
    return out


# one thread, N < 64 queries and n_kv < 512 (or a quantized cache): the reference runs flash_attn_ext_f16_one_chunk, which the oracle restates
@pytest.mark.parametrize("kv_t,D,N,H,Hkv,n_kv,n_past,masked", [
    (O.F16, 64, 1, 4, 2, 1, 0, True), (O.F16, 64, 1, 4, 2, 38, 37, True), (O.F16, 128, 7, 8, 2, 47, 40, True), (O.F16, 128, 1, 4, 4, 300, 299, False),
    (O.F16, 64, 33, 2, 1, 33, 0, True), (O.Q8_0, 128, 1, 8, 2, 600, 599, True), (O.Q8_0, 64, 70, 4, 2, 90, 20, True), (O.Q8_0, 128, 5, 4, 1, 21, 16, False)])
def test_flash_attn_ext_one_chunk_bit_exact(kv_t, D, N, H, Hkv, n_kv, n_past, masked):
    R = O.ref()
    R.ref_set_threads(C.c_int(1))
    try:
        q, k, v, mask = _fa_case(kv_t, D, N, H, Hkv, n_kv, n_past, masked)
        scale = 1.0 / np.sqrt(D)
        ref = np.zeros((N, H, D), np.float32)
        assert R.ref_flash_attn(kv_t, C.c_int64(D), C.c_int64(N), C.c_int64(H), C.c_int64(Hkv), C.c_int64(n_kv), P(q), P(k), P(v), P(mask) if mask is not None else None,
                                C.c_float(scale), P(ref)) == 0
        got = _fa_oracle(kv_t, D, N, H, Hkv, n_kv, q, k, v, mask, scale)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    finally:
        R.ref_set_threads(C.c_int(min(8, __import__("os").cpu_count() or 1)))


# the reference's other two orders: tiled (N >= 64, F16 cache) and split-KV (N == 1, n_kv >= 512, several threads)
# (the fp16 V accumulator of one_chunk loses ~1e-2 of max|out| over 1500 positions; the split-KV order accumulates 8 shorter runs)
@pytest.mark.parametrize("N,n_kv,n_past,threads,tol", [(80, 100, 20, 1, 4e-3), (64, 64, 0, 4, 4e-3), (1, 1500, 1499, 8, 2e-2)])
def test_flash_attn_ext_other_orders_within_tolerance(N, n_kv, n_past, threads, tol):
    R = O.ref()
    R.ref_set_threads(C.c_int(threads))
    try:
        D, H, Hkv = 128, 8, 2
        q, k, v, mask = _fa_case(O.F16, D, N, H, Hkv, n_kv, n_past, True)
        scale = 1.0 / np.sqrt(D)
        ref = np.zeros((N, H, D), np.float32)
        assert R.ref_flash_attn(O.F16, C.c_int64(D), C.c_int64(N), C.c_int64(H), C.c_int64(Hkv), C.c_int64(n_kv), P(q), P(k), P(v), P(mask), C.c_float(scale), P(ref)) == 0
        got = _fa_oracle(O.F16, D, N, H, Hkv, n_kv, q, k, v, mask, scale)
        assert rel_err(got, ref) < tol           # fp16 V accumulation (oracle, one_chunk) vs fp32 tiles / partials
    finally:
        R.ref_set_threads(C.c_int(min(8, __import__("os").cpu_count() or 1)))


@pytest.mark.skipif(not os.path.exists("/root/reference/ggml/src/ggml-common.h"), reason="the reference sources are not on this machine")
def test_packed_codebooks_decode_to_the_reference_grids():
    """chatllm.cpp_amd/csrc/iq_grids.h keeps the IQ1 / IQ2 / IQ3 codebooks as one uint16 per entry (base-3 digits / 3-bit levels); decoded the way iq_grids.h decodes them they
    must be the reference's tables entry for entry (ggml-common.h:528-1615) -- and the sign table must be 7 bits + even parity"""
    import re
    src = open("/root/reference/ggml/src/ggml-common.h").read()
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chatllm.cpp_amd", "csrc", "iq_grids.h")).read()

    def ref_table(name):
        m = re.search(r"GGML_TABLE_BEGIN\(\w+, %s, \w+\)(.*?)GGML_TABLE_END" % name, src, re.S)
        body = m.group(1)
        return [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", body)] or [int(x) for x in re.findall(r"\d+", body)]

    def codes(name):
        m = re.search(r"%s\[\d+\] = \{(.*?)\};" % name, hdr, re.S)
        return [int(x) for x in re.findall(r"\d+", m.group(1))]

    def iq2(code):
        v = 0
        for k in range(8):
            t = (code // 3 ** k) % 3
            v |= (8 + 17 * t + (t >> 1)) << (8 * k)
        return v
    for name, cname in (("iq2xxs_grid", "IQ2XXS_CODE"), ("iq2xs_grid", "IQ2XS_CODE"), ("iq2s_grid", "IQ2S_CODE")):
        assert [iq2(c) for c in codes(cname)] == ref_table(name), name
    assert [sum((62 if ((c >> (3 * k)) & 7) == 7 else 4 + 8 * ((c >> (3 * k)) & 7)) << (8 * k) for k in range(4)) for c in codes("IQ3XXS_CODE")] == ref_table("iq3xxs_grid")
    assert [sum((1 + 2 * ((c >> (3 * k)) & 7)) << (8 * k) for k in range(4)) for c in codes("IQ3S_CODE")] == ref_table("iq3s_grid")
    assert [sum((((c // 3 ** k) % 3 - 1) & 0xff) << (8 * k) for k in range(8)) for c in codes("IQ1S_CODE")] == ref_table("iq1s_grid")
    assert ref_table("ksigns_iq2xs") == [i | ((bin(i).count("1") & 1) << 7) for i in range(128)]
