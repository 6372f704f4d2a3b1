"""GPU parity tests proper: every HIP op, called through the C ABI (ctypes -> libchatllm_hip.so),
against the CPU oracle (oracle/ggml_oracle.c) on the same seeded inputs.

Tiers:  T0 integer block sums and quantized activations: bit-exact;
                        T1 per-op fp32 results: |delta| <= 1e-5 * max|ref| (fp32 summation order differs);
                        byte/index work (set_rows, cpy, get_rows, masks): bit-exact.
"""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from conftest import prefill_mode
from synth_helpers import rand_blocks, rms_boundary_rows

pytestmark = pytest.mark.gpu
rng = np.random.default_rng(11)
T1 = 1e-5


def rel_err(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) / (np.max(np.abs(b)) + 1e-30))


def P(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- T0: activation quantizers ---------------------------------------------------------------
@pytest.mark.parametrize("K", [32, 256, 4096, 14336, 29568])
def test_quantize_q8_0_bit_exact(gpu, K):
    T = gpu.Tensor
    for scale in (1.0, 1e-3, 300.0):
        x = (rng.standard_normal(K) * scale).astype(np.float32)
        if K >= 256:
            x[:32] = 0.0
            x[40] = x[41] = -x[42]
            x[64:96] = np.arange(32, dtype=np.float32) + 0.5
            x[96:128] = (np.arange(32, dtype=np.float32) - 16) * 0.5
            x[96] = 127.0                                  # id == 1: every k + 0.5 is a rounding tie
        dx = T.from_numpy(x)
        dy = T(gpu.I32, [K // 32 * 34 // 2 + 8])           # raw bytes
        gpu.lib.check(gpu.lib.get().cllm_quantize_row_q8_0(None, dx.data_ptr(), dy.data_ptr(), K), "q8_0")
        got = dy.raw()[: K // 32 * 34]
        assert np.array_equal(got, O.quantize_q8_0(x))


@pytest.mark.parametrize("K", [32, 256, 4096, 14336])
def test_quantize_q8_1_bit_exact(gpu, K):
    """the activation format of Q4_1 weights (quantize_row_q8_1, x86 branch): Q8_0's quants + s = fp16(d * sum q)"""
    T = gpu.Tensor
    for scale in (1.0, 1e-3, 300.0, 6e4):
        x = (rng.standard_normal(K) * scale).astype(np.float32)
        if K >= 256:
            x[:32] = 0.0
            x[64:96] = np.arange(32, dtype=np.float32) + 0.5
            x[96:128] = np.abs(x[96:128])
        dx = T.from_numpy(x)
        dy = T(gpu.I32, [K // 32 * 36 // 4 + 8])
        gpu.lib.check(gpu.lib.get().cllm_quantize_row_q8_1(None, dx.data_ptr(), dy.data_ptr(), K), "q8_1")
        got = dy.raw()[: K // 32 * 36]
        assert np.array_equal(got, O.quantize_q8_1(x))


@pytest.mark.parametrize("K", [256, 4096, 14336])
def test_quantize_q8_K_bit_exact(gpu, K):
    T = gpu.Tensor
    for scale in (1.0, 1e-4, 50.0):
        x = (rng.standard_normal(K) * scale).astype(np.float32)
        x[10] = -x[3]
        x[3] = np.float32(np.max(np.abs(x[:256])) * 2)     # the max sits at index 3, an equal-magnitude negative at 10
        x[10] = -x[3]
        x[256 * (K // 256 - 1):] = 0.0                     # an all-zero super-block
        dx = T.from_numpy(x)
        dy = T(gpu.I32, [K // 256 * 292 // 4 + 8])
        gpu.lib.check(gpu.lib.get().cllm_quantize_row_q8_K(None, dx.data_ptr(), dy.data_ptr(), K), "q8_K")
        got = dy.raw()[: K // 256 * 292]
        assert np.array_equal(got, O.quantize_q8_K(x))


@pytest.mark.parametrize("t", [O.Q4_0, O.Q4_1, O.Q8_0, O.Q4_K])
def test_block_integer_sums_bit_exact(gpu, t):
    K = 4096
    w = rand_blocks(t, 1, K, rng)
    x = rng.standard_normal(K).astype(np.float32)
    a = O.quantize_q8_K(x) if t == O.Q4_K else O.quantize_q8_1(x) if t == O.Q4_1 else O.quantize_q8_0(x)
    _, want = O.vec_dot(t, K, w, a)
    dw = gpu.Tensor.from_numpy(w, t, [K, 1])
    dx = gpu.Tensor.from_numpy(x)
    out = gpu.Tensor(gpu.I32, [want.size])
    gpu.lib.check(gpu.lib.get().cllm_vec_dot_isums(None, t, K, dw.data_ptr(), dx.data_ptr(), out.data_ptr()), "isums")
    assert np.array_equal(out.numpy().ravel(), want)


# ---- mul_mat -----------------------------------------------------------------------------------
def _mm_case(gpu, t, K, N, M, ne02=1, ne12=1):
    if t in (O.F16, O.F32):
        w = rng.standard_normal((ne02, N, K)).astype(O.NP_OF[t])
    else:
        w = rand_blocks(t, N * ne02, K, rng)
    x = rng.standard_normal((ne12, M, K)).astype(np.float32)
    want = np.zeros((ne12, M, N), np.float32)
    O.mul_mat(O.tensor(w, t, [K, N, ne02]), O.tensor(x, O.F32, [K, M, ne12]), O.tensor(want, O.F32, [N, M, ne12]))
    dw = gpu.Tensor.from_numpy(w, t, [K, N, ne02])
    dx = gpu.Tensor.from_numpy(x)
    got = gpu.ops.mul_mat(dw, dx).numpy()
    return got, want


@pytest.mark.parametrize("t", [O.Q4_K, O.Q4_0, O.Q8_0, O.Q4_1])
@pytest.mark.parametrize("K,N,M", [(256, 1, 1), (512, 7, 1), (4096, 130, 1), (2048, 64, 2), (1024, 33, 3), (768, 40, 4), (512, 20, 5), (1280, 24, 8),
                                   # one column, even row counts: the 32-block types take the rows-side-by-side kernel (gemv_rows32.hip): whole and ragged steps, many units per wave
                                   (4096, 128, 1), (512, 64, 1), (1024, 8, 1), (2112, 24, 1), (4096, 8192, 1), (11008, 256, 1)])
def test_mul_mat_quant_gemv(gpu, t, K, N, M):
    """1..8 columns: the mat-vec kernels accumulate in the reference's AVX2 order (q4k.h / q32.h) -- bit-identical to the oracle == libggml-cpu.so"""
    if K % O.BLCK[t]:
        pytest.skip("K is not a multiple of the type's block")
    got, want = _mm_case(gpu, t, K, N, M)
    assert np.array_equal(got.view(np.uint32).ravel(), want.view(np.uint32).ravel()), rel_err(got, want)


@pytest.fixture()
def rows32_mode(gpu):
    """gemv_rows32.hip's form: 0 = k_gemv_dec takes the launch, 2 / 4 / 8 = that many rows per wave (8: prologue-1 launches only), 1 = the launcher's pick"""
    L = gpu.lib.get()

    def set_mode(m):
        L.cllm_debug_set_gemv_rows32(int(m))
    yield set_mode
    L.cllm_debug_set_gemv_rows32(1)


@pytest.mark.parametrize("t", [O.Q4_0, O.Q8_0, O.Q4_1])
@pytest.mark.parametrize("mode", [0, 2, 4])
@pytest.mark.parametrize("K,N", [(64, 4), (512, 64), (2112, 24), (4096, 1028), (14336, 96), (29568, 40), (1024, 20000)])
def test_mul_mat_quant_gemv_every_rows32_form(gpu, rows32_mode, t, mode, K, N):
    """the one-column mat-vec of the 32-block types through each form of the decode kernel (one row per wave; 2 / 4 rows side by side): every output word is the oracle's"""
    rows32_mode(mode)
    got, want = _mm_case(gpu, t, K, N, 1)
    assert np.array_equal(got.view(np.uint32).ravel(), want.view(np.uint32).ravel()), rel_err(got, want)


@pytest.mark.parametrize("t", [O.Q4_0, O.Q8_0, O.Q4_1])
@pytest.mark.parametrize("mode", [2, 4, 8])
@pytest.mark.parametrize("K,N,epi", [(4096, 384, 0), (4096, 4112, 1), (14336, 64, 0), (512, 18000, 1), (1056, 200, 0)])
def test_gemv_rows32_prologue_1_forms_equal_the_node_sequence(gpu, rows32_mode, t, mode, K, N, epi):
    """RMS_NORM * weight -> quantize -> mat-vec (+ bias + residual | SiLU(gate) * up) in one launch, 2 / 4 / 8 rows per wave == the node sequence through k_gemv_dec"""
    ops, T, L = gpu.ops, gpu.Tensor, gpu.lib.get()
    w = T.from_numpy(rand_blocks(t, N, K, rng), t, [K, N])
    x = T.from_numpy(rng.standard_normal((1, K)).astype(np.float32))
    g = T.from_numpy((1 + 0.1 * rng.standard_normal(K)).astype(np.float32))
    r = T.from_numpy(rng.standard_normal((1, N)).astype(np.float32))
    rows32_mode(0)
    y = ops.mul_mat(w, ops.rms_norm_mul(x, g, 1e-5))
    if epi:
        yn = y.numpy().reshape(N // 2, 2)
        gate, up = T.from_numpy(np.ascontiguousarray(yn[:, 0])), T.from_numpy(np.ascontiguousarray(yn[:, 1]))
        want = ops.mul(ops.silu(gate), up).numpy().reshape(-1)
    else:
        want = ops.add(y, r).numpy().reshape(-1)
    rows32_mode(mode)
    out = T(gpu.F32, [N // 2 if epi else N, 1])
    cw = w.c()
    gpu.lib.check(L.cllm_op_mul_mat_vec_fused(None, C.byref(cw), 1, x.data_ptr(), g.data_ptr(), 1e-5, epi, None if epi else r.data_ptr(), out.data_ptr()), "fused")
    assert np.array_equal(out.numpy().reshape(-1).view(np.uint32), want.view(np.uint32))


@pytest.fixture()
def team32_mode(gpu):
    """gemv_team32.hip: 0 = off, 4 / 5 / 8 / 16 = that many waves per team of 8 rows, 1 = the launcher's pick (few rows per CU only)"""
    L = gpu.lib.get()

    def set_mode(m):
        L.cllm_debug_set_gemv_team32(int(m))
    yield set_mode
    L.cllm_debug_set_gemv_team32(1)
    assert L.cllm_debug_gemv_team32_error() == 0, "a hand-off between the waves of a team timed out"


@pytest.mark.parametrize("t", [O.Q4_0, O.Q8_0, O.Q4_1])
@pytest.mark.parametrize("team", [1, 4, 5, 8, 16])
@pytest.mark.parametrize("K,N", [(256, 8), (512, 64), (2112, 24), (4096, 1032), (4096, 4096), (14336, 96), (29568, 40), (1024, 20000)])
def test_mul_mat_quant_gemv_every_team32_form(gpu, rows32_mode, team32_mode, t, team, K, N):
    """the one-column mat-vec of the 32-block types through the team kernel (4 / 8 / 16 waves share 8 rows: emit waves -> LDS -> one chain wave): every output word is the oracle's;
    one unit, ragged last steps, several units per team, more workgroups' worth of units than CUs"""
    rows32_mode(0)
    team32_mode(team)
    got, want = _mm_case(gpu, t, K, N, 1)
    assert np.array_equal(got.view(np.uint32).ravel(), want.view(np.uint32).ravel()), rel_err(got, want)


@pytest.mark.parametrize("t", [O.Q4_0, O.Q8_0, O.Q4_1])
@pytest.mark.parametrize("team", [1, 4, 5, 8, 16])
@pytest.mark.parametrize("K,N,pro", [(4096, 4096, 1), (4096, 6144, 1), (14336, 4096, 4), (4096, 1024, 2), (8192, 8192, 1), (28672, 1024, 4), (1056, 200, 1), (512, 18000, 4)])
def test_gemv_team32_fused_forms_equal_the_node_sequence(gpu, rows32_mode, team32_mode, t, team, K, N, pro):
    """[RMS_NORM * weight | SiLU(gate) * up |] quantize -> mat-vec + residual in one launch of the team kernel == the node sequence through k_gemv_dec (the decode step's qkv / o / down shapes)"""
    ops, T, L = gpu.ops, gpu.Tensor, gpu.lib.get()
    w = T.from_numpy(rand_blocks(t, N, K, rng), t, [K, N])
    x = T.from_numpy(rng.standard_normal((1, K)).astype(np.float32))
    g = T.from_numpy((1 + 0.1 * rng.standard_normal((1, K))).astype(np.float32))
    r = T.from_numpy(rng.standard_normal((1, N)).astype(np.float32))
    rows32_mode(0)
    team32_mode(0)
    act = ops.rms_norm_mul(x, T.from_numpy(g.numpy().reshape(K)), 1e-5) if pro == 1 else ops.silu_mul(x, g) if pro == 4 else x
    want = ops.add(ops.mul_mat(w, act), r).numpy().reshape(-1)
    team32_mode(team)
    out = T(gpu.F32, [N, 1])
    cw = w.c()
    gpu.lib.check(L.cllm_op_mul_mat_vec_fused(None, C.byref(cw), pro, x.data_ptr(), g.data_ptr() if pro != 2 else None, 1e-5, 0, r.data_ptr(), out.data_ptr()), "fused")
    assert np.array_equal(out.numpy().reshape(-1).view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("t", [O.Q5_K, O.Q6_K])
@pytest.mark.parametrize("K,N,M,ne02,ne12", [(256, 1, 1, 1, 1), (512, 7, 1, 1, 1), (4096, 130, 1, 1, 1), (2048, 64, 2, 1, 1), (1024, 33, 5, 1, 1), (1280, 24, 8, 1, 1), (768, 40, 9, 1, 1),
                                             (512, 19, 40, 1, 1), (256, 12, 3, 2, 4), (14336, 48, 1, 1, 1)])
def test_mul_mat_k_quants_any_columns_bit_exact(gpu, t, K, N, M, ne02, ne12):
    """Q5_K / Q6_K (gemv_kq.hip): 8 lanes per row, lane = AVX lane, serial fma chain in a register -- bit-identical to libggml-cpu.so for every column count"""
    got, want = _mm_case(gpu, t, K, N, M, ne02, ne12)
    assert np.array_equal(got.view(np.uint32).ravel(), want.view(np.uint32).ravel()), rel_err(got, want)


@pytest.mark.parametrize("t", [O.Q5_0, O.Q5_1, O.IQ4_NL, O.MXFP4, O.Q2_K, O.Q3_K, O.IQ4_XS, O.TQ1_0, O.TQ2_0, O.IQ2_XXS, O.IQ2_XS, O.IQ2_S, O.IQ3_XXS, O.IQ3_S, O.IQ1_S, O.IQ1_M])
@pytest.mark.parametrize("K,N,M,ne02,ne12", [(256, 1, 1, 1, 1), (512, 7, 1, 1, 1), (4096, 130, 1, 1, 1), (2048, 64, 2, 1, 1), (1024, 33, 5, 1, 1), (1280, 24, 8, 1, 1), (768, 40, 9, 1, 1),
                                             (512, 19, 40, 1, 1), (256, 12, 3, 2, 4), (14336, 48, 1, 1, 1), (96, 10, 1, 1, 1), (96, 10, 3, 1, 1), (2080, 9, 1, 1, 1)])
def test_mul_mat_other_formats_any_columns_bit_exact(gpu, t, K, N, M, ne02, ne12):
    """Q5_0 / Q5_1 / IQ4_NL / MXFP4 / Q2_K / Q3_K (gemv_kq.hip): the formats other stock model files carry, every column count, bit-identical to libggml-cpu.so
    -- including the codebook formats' paired accumulators with an unpaired last block (K = 96, 2080) and IQ4_NL's other order for >= 2 columns (tinyBLAS)"""
    if K % O.BLCK[t]:
        pytest.skip("K is not a multiple of the type's block")
    got, want = _mm_case(gpu, t, K, N, M, ne02, ne12)
    assert np.array_equal(got.view(np.uint32).ravel(), want.view(np.uint32).ravel()), rel_err(got, want)


@pytest.mark.parametrize("t,K", [(O.Q4_K, 14336), (O.Q4_0, 14336), (O.Q8_0, 29568), (O.Q4_K, 8192), (O.Q4_1, 14336), (O.Q4_1, 29568)])
def test_mul_mat_quant_long_rows(gpu, t, K):
    got, want = _mm_case(gpu, t, K, 96, 1)
    assert np.array_equal(got.view(np.uint32).ravel(), want.view(np.uint32).ravel()), rel_err(got, want)


@pytest.mark.parametrize("t", [O.Q4_K, O.Q4_0, O.Q8_0, O.Q4_1])
@pytest.mark.parametrize("K,N,M", [(512, 64, 16), (1024, 100, 33), (4096, 256, 128), (256, 17, 9),
                                   (768, 130, 70), (4352, 300, 257)])        # two token tiles, ragged N / M / K
def test_mul_mat_quant_gemm(gpu, t, K, N, M):
    """any number of columns, bit-identical to the oracle == libggml-cpu.so: up to 32 columns the exact-order mat-vec in column chunks; beyond, mmx.hip
    (integer block sums on the K = 4 matrix-core instruction, the fp32 chains in the reference's block order)"""
    got, want = _mm_case(gpu, t, K, N, M)
    assert np.array_equal(got.view(np.uint32).ravel(), want.view(np.uint32).ravel()), rel_err(got, want)


@pytest.mark.parametrize("t", [O.Q4_K, O.Q4_0, O.Q8_0, O.Q4_1])
@pytest.mark.parametrize("K,N,M", [(14336, 70, 40), (4128, 65, 33), (256, 5, 100), (2048, 129, 513), (32, 3, 64),
                                   # a prompt-sized product (>= 256 tokens, >= 1024 rows): Q4_0 / Q8_0 stage the activations from the once-converted fp16 copy
                                   (512, 1056, 300), (1056, 1024, 257)])
def test_mul_mat_quant_exact_many_columns(gpu, t, K, N, M):
    """mmx.hip at its edges: long rows (56 super-blocks), K that is not a multiple of the 256-element stage (32-block types), one-block rows,
    ragged row / token tiles, several token tiles -- every output word equals the reference's"""
    if t == O.Q4_K and K % 256:
        pytest.skip("Q4_K rows are whole super-blocks")
    got, want = _mm_case(gpu, t, K, N, M)
    assert np.array_equal(got.view(np.uint32).ravel(), want.view(np.uint32).ravel()), rel_err(got, want)


@pytest.mark.parametrize("t", [O.Q4_K, O.Q4_0, O.Q8_0, O.Q4_1])
@pytest.mark.parametrize("K,N,M", [(1024, 100, 33), (4096, 256, 128), (768, 130, 70), (4352, 300, 257)])
def test_mul_mat_quant_gemm_fast_mode(gpu, t, K, N, M):
    """prefill mode 0 (CLLM_PREFILL=fast): the int8-MFMA GEMM (mmq.hip): exact integer block sums, its own fp32 summation order (tier T1)"""
    with prefill_mode(gpu, 0):
        got, want = _mm_case(gpu, t, K, N, M)
    assert rel_err(got, want) < T1


@pytest.mark.parametrize("t", [O.Q4_K, O.Q4_0, O.Q8_0, O.Q4_1])
@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("K,N,M", [(512, 64, 40), (4096, 256, 128), (768, 130, 70), (4352, 300, 257), (14336, 140, 33), (1024, 520, 513)])
def test_mul_mat_quant_dense_f16_mode(gpu, t, tile, K, N, M):
    """CLLM_PREFILL=f16 (opt-in; north star: block dequant staged through LDS + fp16 MFMA tiles, dense_f16.hip): the weights are dequantized to fp16 inside the
    GEMM's staging, the activations rounded to fp16 -- NOT the reference's computation (no activation quantization), so it is checked against the oracle's
    dequantize_row_* values times the fp16-rounded activations in float64: |delta| <= 1.5e-3 * max|ref| (the fp16 rounding of the weights: 2^-11 per element)"""
    import ctypes as C2
    lib = C2.CDLL(gpu.lib.SO_PATH)
    w = rand_blocks(t, N, K, rng)
    x = rng.standard_normal((1, M, K)).astype(np.float32)
    wd = np.stack([O.dequantize(t, w[r], K) for r in range(N)]).astype(np.float64)
    want = x[0].astype(np.float16).astype(np.float64) @ wd.T                   # [M, N]
    lib.cllm_debug_set_prefill_f16(1)
    lib.cllm_debug_set_mmd_tile(tile)                                           # both workgroup tiles (128 x 128, 256 x 256 tokens x rows), whole and ragged
    try:
        with prefill_mode(gpu, 0):
            got = gpu.ops.mul_mat(gpu.Tensor.from_numpy(w, t, [K, N]), gpu.Tensor.from_numpy(x)).numpy().reshape(M, N)
    finally:
        lib.cllm_debug_set_prefill_f16(0)
        lib.cllm_debug_set_mmd_tile(0)
    assert np.all(np.isfinite(got))
    assert float(np.max(np.abs(got - want))) <= 1.5e-3 * float(np.max(np.abs(want))), float(np.max(np.abs(got - want)) / np.max(np.abs(want)))


@pytest.mark.parametrize("t", [O.F16, O.F32])
@pytest.mark.parametrize("K,N,M,ne02,ne12", [(128, 50, 1, 1, 1), (128, 37, 3, 2, 8), (64, 9, 5, 1, 4), (100, 11, 2, 1, 1), (1031, 16, 1, 2, 2),
                                             # >= 32 columns: the F16 case runs on the matrix cores (mma_f16.hip): full tiles, ragged N / M / K, GQA broadcast
                                             (128, 256, 128, 1, 1), (128, 200, 77, 2, 8), (328, 130, 33, 1, 4), (72, 17, 40, 1, 1), (8, 3, 32, 1, 2)])
def test_mul_mat_float(gpu, t, K, N, M, ne02, ne12):
    """ggml_vec_dot_f16 / _f32's order, or tinyBLAS<8>'s where the reference takes it -- bit-identical for every column count (beyond 32 columns the F16
    contraction runs as fmaf chains on the f32 matrix cores, mmf_exact.hip); prefill mode 0: the fp16 MFMA kernel (tier T1)"""
    got, want = _mm_case(gpu, t, K, N, M, ne02, ne12)
    assert np.array_equal(got.view(np.uint32).ravel(), want.view(np.uint32).ravel()), rel_err(got, want)
    if M > 32 and t == O.F16:
        with prefill_mode(gpu, 0):
            got, want = _mm_case(gpu, t, K, N, M, ne02, ne12)
        assert rel_err(got, want) < T1


@pytest.mark.parametrize("K,N,M,ne02,ne12", [(128, 300, 200, 2, 8), (128, 301, 70, 1, 4), (200, 128, 64, 2, 2), (203, 64, 40, 1, 1), (1000, 128, 100, 1, 2),
                                             (1029, 128, 33, 2, 4), (24, 8, 64, 1, 1), (31, 5, 64, 1, 1), (4096, 128, 64, 1, 1)])
def test_mul_mat_f16_exact_many_columns(gpu, K, N, M, ne02, ne12):
    """mmf_exact.hip in both of the reference's orders: tinyBLAS<8> (K % 8 == 0, rows % 4 == 0) and ggml_vec_dot_f16 (otherwise: 32 accumulators, leftovers in
    double), K.Q-like (K = 128, many rows) and V.P-like (K = positions) shapes, ragged tiles, K below one chain step, GQA broadcast"""
    got, want = _mm_case(gpu, O.F16, K, N, M, ne02, ne12)
    assert np.array_equal(got.view(np.uint32).ravel(), want.view(np.uint32).ravel()), rel_err(got, want)


def test_mul_mat_rejects_bad_arguments(gpu):
    Tn = gpu.Tensor
    w = Tn.from_numpy(rand_blocks(O.Q4_K, 4, 256, rng), O.Q4_K, [256, 4])
    x = Tn.from_numpy(np.zeros((1, 512), np.float32))
    with pytest.raises(gpu.lib.CllmError):
        gpu.ops.mul_mat(w, x)                       # K mismatch
    x16 = Tn.from_numpy(np.zeros((1, 256), np.float16))
    with pytest.raises(gpu.lib.CllmError):
        gpu.ops.mul_mat(w, x16)                     # src1 must be F32


def test_mul_mat_empty(gpu):
    w = gpu.Tensor.from_numpy(rand_blocks(O.Q8_0, 4, 64, rng), O.Q8_0, [64, 4])
    x = gpu.Tensor(gpu.F32, [64, 0])
    out = gpu.ops.mul_mat(w, x)
    assert out.ne[:2] == [4, 0]


@pytest.mark.parametrize("t", [O.Q4_K, O.Q8_0, O.Q4_0, O.Q4_1, O.Q5_K, O.Q6_K, O.Q2_K, O.Q3_K, O.Q5_0, O.Q5_1, O.IQ4_NL, O.MXFP4, O.IQ4_XS, O.TQ1_0, O.TQ2_0, O.IQ2_XXS, O.IQ2_XS, O.IQ2_S, O.IQ3_XXS, O.IQ3_S, O.IQ1_S, O.IQ1_M])
def test_mul_mat_id(gpu, t):
    """expert weights of every quantized type (Q4_K_M-style Mixtral files keep expert tensors in Q5_K / Q6_K; GPT-OSS experts are MXFP4): the tuned types through
    mmvq / the decode mat-vec, the coverage types one grid slice per (token, slot) of gemv_kq.hip -- always vec_dot's one-column order"""
    K, N, E, U, Tk = 512, 40, 8, 2, 3
    w = rand_blocks(t, N * E, K, rng)
    for nb1 in (1, U):
        x = rng.standard_normal((Tk, nb1, K)).astype(np.float32)
        ids = rng.integers(0, E, (Tk, U)).astype(np.int32)
        want = np.zeros((Tk, U, N), np.float32)
        O.mul_mat_id(O.tensor(w, t, [K, N, E]), O.tensor(x, O.F32, [K, nb1, Tk]), O.tensor(ids, O.I32, [U, Tk]), O.tensor(want, O.F32, [N, U, Tk]))
        got = gpu.ops.mul_mat_id(gpu.Tensor.from_numpy(w, t, [K, N, E]), gpu.Tensor.from_numpy(x), gpu.Tensor.from_numpy(ids)).numpy()
        assert np.array_equal(got.reshape(want.shape).view(np.uint32), want.view(np.uint32)), rel_err(got.reshape(want.shape), want)      # the reference's AVX2 order


# ---- norm / rope / softmax / elementwise -----------------------------------------------------------
@pytest.mark.parametrize("n0,rows", [(8, 3), (100, 2), (4096, 5), (8192, 1)])
def test_rms_norm(gpu, n0, rows):
    x = rng.standard_normal((rows, n0)).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(n0)).astype(np.float32)
    want = np.zeros_like(x)
    O.rms_norm(O.tensor(x, O.F32, [n0, rows]), O.tensor(want, O.F32, [n0, rows]), 1e-5)
    dx = gpu.Tensor.from_numpy(x)
    got = gpu.ops.rms_norm(dx, 1e-5).numpy().reshape(x.shape)
    # the workgroup adds the squares as a tree, the CPU serially: rms_scale (common.h) proves per row that the float mean cannot depend on the order, or redoes
    # the sum in the reference's order -- every word is the oracle's (test_rms_norm_rows_on_a_rounding_boundary: the rows where the order decides)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    want2 = want * w                                 # the MUL node: one more rounding
    got2 = gpu.ops.rms_norm_mul(dx, gpu.Tensor.from_numpy(w), 1e-5).numpy().reshape(x.shape)
    assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32))


@pytest.mark.parametrize("n0", [4096, 1000, 8192, 14336])
def test_rms_norm_rows_on_a_rounding_boundary(gpu, n0):
    """rows whose mean of squares lies on a float rounding boundary (tests/synth_helpers.rms_boundary_rows): the float the reference rounds it to depends on the
    ORDER of its serial double accumulation (ops.cpp:3736-3741); a tree sum gets about half of them wrong by one ulp of the mean.  Every word must be the oracle's."""
    rows = 24
    x = rms_boundary_rows(n0, rows, np.random.default_rng(n0))
    order_matters = 0
    for r in x:                                      # the test is only worth something if the order does decide: a pairwise sum must disagree with the serial one
        s = (r * r).astype(np.float64)
        ser = 0.0
        for v in s:
            ser += v
        order_matters += np.float32(ser / n0) != np.float32(float(np.sum(s)) / n0)
    assert order_matters >= 4, order_matters
    w = (1 + 0.1 * rng.standard_normal(n0)).astype(np.float32)
    want = np.zeros_like(x)
    O.rms_norm(O.tensor(x, O.F32, [n0, rows]), O.tensor(want, O.F32, [n0, rows]), 1e-5)
    dx = gpu.Tensor.from_numpy(x)
    got = gpu.ops.rms_norm(dx, 1e-5).numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int(np.sum(got.view(np.uint32) != want.view(np.uint32)))
    got2 = gpu.ops.rms_norm_mul(dx, gpu.Tensor.from_numpy(w), 1e-5).numpy()
    assert np.array_equal(got2.view(np.uint32), (want * w).view(np.uint32))
    # in place (the serial pass reads the row before any thread stores)
    gpu.ops.rms_norm(dx, 1e-5, dst=dx)
    assert np.array_equal(dx.numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("t,K,N", [(O.Q4_K, 4096, 512), (O.Q4_K, 8192, 256), (O.Q4_0, 4096, 4096), (O.Q8_0, 4096, 384), (O.Q4_1, 4096, 200), (O.Q4_0, 4096, 28672)])
def test_norm_prologues_on_a_rounding_boundary(gpu, t, K, N):
    """the same rows through the RMS_NORM prologues of the decode mat-vecs (k_gemv_dec, k_gemv_rows32, k_gemv_team32) and of the prompt's quantizer
    (k_rms_norm_quantize): equal to the oracle's RMS_NORM -> MUL -> MUL_MAT, bit for bit"""
    ops, T, L = gpu.ops, gpu.Tensor, gpu.lib.get()
    cols = 12
    x = rms_boundary_rows(K, cols, np.random.default_rng(K + N))
    g = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    wq = rand_blocks(t, N, K, rng)
    xn = np.zeros_like(x)
    O.rms_norm(O.tensor(x, O.F32, [K, cols]), O.tensor(xn, O.F32, [K, cols]), 1e-5)
    xn = xn * g
    want = np.zeros((cols, N), np.float32)
    O.mul_mat(O.tensor(wq, t, [K, N]), O.tensor(xn, O.F32, [K, cols]), O.tensor(want, O.F32, [N, cols]))
    w = T.from_numpy(wq, t, [K, N])
    dg = T.from_numpy(g.reshape(1, K))
    cw = w.c()
    for c in range(min(cols, 6)):                    # one token: the decode kernels' prologue
        out = T(gpu.F32, [N, 1])
        dx = T.from_numpy(x[c:c + 1])
        gpu.lib.check(L.cllm_op_mul_mat_vec_fused(None, C.byref(cw), 1, dx.data_ptr(), dg.data_ptr(), 1e-5, 0, None, out.data_ptr()), "fused")
        assert np.array_equal(out.numpy().reshape(N).view(np.uint32), want[c].view(np.uint32)), c
    if N <= 4096:
        got = ops.mul_mat_ex(w, T.from_numpy(x), pro=1, norm_w=T.from_numpy(g), eps=1e-5).numpy()      # a prompt: norm inside the quantizer
        assert np.array_equal(got.reshape(want.shape).view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("mode,hd,n_dims,ff", [(0, 128, 128, False), (2, 128, 128, False), (0, 64, 32, False), (2, 64, 64, True)])
def test_rope(gpu, mode, hd, n_dims, ff):
    heads, qlen = 5, 6
    x = rng.standard_normal((qlen, heads, hd)).astype(np.float32)
    pos = np.array([0, 1, 7, 100, 1000, 4095], np.int32)
    ffv = (1.0 + rng.random(n_dims // 2)).astype(np.float32) if ff else None
    want = np.zeros_like(x)
    O.rope(O.tensor(x, O.F32, [hd, heads, qlen]), pos, ffv, O.tensor(want, O.F32, [hd, heads, qlen]), n_dims, mode, 500000.0)
    got = gpu.ops.rope_ext(gpu.Tensor.from_numpy(x), gpu.Tensor.from_numpy(pos), gpu.Tensor.from_numpy(ffv) if ff else None,
                           n_dims, mode, 0, 500000.0).numpy()
    # theta is bit-identical (iterated product), cos / sin are glibc's to the bit (glibc_math.h), the rotation is the reference build's fma form
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), float(np.max(np.abs(got - want)))
    # in place (rope_ext_inplace) gives the same bytes
    dxi = gpu.Tensor.from_numpy(x)
    gpu.ops.rope_ext(dxi, gpu.Tensor.from_numpy(pos), gpu.Tensor.from_numpy(ffv) if ff else None, n_dims, mode, 0, 500000.0, inplace=True)
    assert np.array_equal(dxi.numpy(), got)


def test_rope_yarn(gpu):
    hd, heads, qlen = 64, 2, 4
    x = rng.standard_normal((qlen, heads, hd)).astype(np.float32)
    pos = np.array([0, 3, 17, 50, 333, 1000, 4095, 20000], np.int32)
    qlen = len(pos)
    x = rng.standard_normal((qlen, heads, hd)).astype(np.float32)
    want = np.zeros_like(x)
    kw = dict(n_ctx_orig=4096, freq_scale=0.25, ext_factor=1.0, attn_factor=1.2, beta_fast=32.0, beta_slow=1.0)
    O.rope(O.tensor(x, O.F32, [hd, heads, qlen]), pos, None, O.tensor(want, O.F32, [hd, heads, qlen]), hd, 2, 10000.0, **kw)
    got = gpu.ops.rope_ext(gpu.Tensor.from_numpy(x), gpu.Tensor.from_numpy(pos), None, hd, 2, freq_base=10000.0, **kw).numpy()
    # theta's YaRN mix and the magnitude correction are the reference build's two fmas, logf is the host's libm, cos / sin glibc's to the bit: every word equal
    assert np.array_equal(got.reshape(want.shape).view(np.uint32), want.view(np.uint32)), float(np.max(np.abs(got.reshape(want.shape) - want)))


@pytest.mark.parametrize("n0", [1, 5, 8, 33, 1024, 4097])
def test_soft_max(gpu, n0):
    x = (rng.standard_normal((2, 3, n0)) * 3).astype(np.float32)
    want = np.zeros_like(x)
    O.soft_max(O.tensor(x, O.F32, [n0, 3, 2]), None, O.tensor(want, O.F32, [n0, 3, 2]))
    got = gpu.ops.soft_max(gpu.Tensor.from_numpy(x)).numpy()
    # the CPU's AVX2 body (same polynomial, same group sums) and its n mod 8 tail (glibc's expf, glibc_math.h): bit-identical
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), float(np.max(np.abs(got - want)))


def test_soft_max_ext_mask(gpu):
    n0, n1, n2 = 40, 6, 3
    x = rng.standard_normal((n2, n1, n0)).astype(np.float32)
    mask = np.where(rng.random((n1, n0)) < 0.3, -np.inf, 0.0).astype(np.float32)
    mask[:, 0] = 0.0
    for f16 in (False, True):
        mk = mask.astype(np.float16) if f16 else mask
        want = np.zeros_like(x)
        O.soft_max(O.tensor(x, O.F32, [n0, n1, n2]), O.tensor(mk, O.F16 if f16 else O.F32, [n0, n1]), O.tensor(want, O.F32, [n0, n1, n2]), scale=0.125)
        got = gpu.ops.soft_max_ext(gpu.Tensor.from_numpy(x), gpu.Tensor.from_numpy(mk), 0.125, 0.0).numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), float(np.max(np.abs(got - want)))
        assert np.array_equal(got == 0, want == 0)


@pytest.mark.parametrize("qlen,n_past", [(1, 0), (1, 300), (7, 0), (5, 13),
                                         (520, 0), (24, 1000), (40, 4400),         # rows of >= 512 (multiple of 8): register-resident kernel
                                         (24, 9000), (8, 20000)])                  # rows > 8192: one workgroup per row, row in LDS
def test_scale_mask_soft_max_equals_three_nodes(gpu, qlen, n_past):
    nh, n_kv = 4, n_past + qlen
    x = rng.standard_normal((nh, qlen, n_kv)).astype(np.float32)
    s = 1.0 / np.sqrt(128.0)
    want = np.zeros_like(x)
    St = O.tensor(want, O.F32, [n_kv, qlen, nh])
    O.scale(O.tensor(x, O.F32, [n_kv, qlen, nh]), St, s)
    O.diag_mask_inf(St, St, n_past)
    O.soft_max(St, None, St)
    dx = gpu.Tensor.from_numpy(x)
    fused = gpu.ops.scale_mask_soft_max(dx, s, n_past).numpy()
    chain = gpu.ops.soft_max(gpu.ops.diag_mask_inf(gpu.ops.scale(dx, s), n_past)).numpy()
    assert np.array_equal(fused, chain)               # fusion does not change a bit
    assert np.array_equal(fused.view(np.uint32), want.view(np.uint32)), float(np.max(np.abs(fused - want)))
    assert np.array_equal(fused == 0, want == 0)      # the causal structure is exact


def test_diag_mask_scale_add_mul_silu(gpu):
    x = (rng.standard_normal((2, 5, 61)) * 4).astype(np.float32)
    b = rng.standard_normal((1, 1, 61)).astype(np.float32)
    dx, db = gpu.Tensor.from_numpy(x), gpu.Tensor.from_numpy(b)
    want = np.zeros_like(x)
    Xo, Wo, Bo = O.tensor(x, O.F32, [61, 5, 2]), O.tensor(want, O.F32, [61, 5, 2]), O.tensor(b, O.F32, [61, 1, 1])
    O.diag_mask_inf(Xo, Wo, 3)
    assert np.array_equal(gpu.ops.diag_mask_inf(dx, 3).numpy().view(np.uint32), want.view(np.uint32))
    O.scale(Xo, Wo, 0.3)
    assert np.array_equal(gpu.ops.scale(dx, 0.3).numpy().view(np.uint32), want.view(np.uint32))
    O.add(Xo, Bo, Wo)
    assert np.array_equal(gpu.ops.add(dx, db).numpy().view(np.uint32), want.view(np.uint32))
    O.mul(Xo, Bo, Wo)
    assert np.array_equal(gpu.ops.mul(dx, db).numpy().view(np.uint32), want.view(np.uint32))
    O.silu(Xo, Wo)
    got = gpu.ops.silu(dx).numpy()
    nv = 61 & ~7
    assert np.array_equal(got[..., :nv].view(np.uint32), want[..., :nv].view(np.uint32))   # polynomial body: bit-exact
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))                       # and the libm tail (glibc's expf)
    # fused silu(g)*u == the two nodes
    u = rng.standard_normal(x.shape).astype(np.float32)
    du = gpu.Tensor.from_numpy(u)
    assert np.array_equal(gpu.ops.silu_mul(dx, du).numpy(), gpu.ops.mul(gpu.ops.silu(dx), du).numpy())


# ---- KV-concat: set_rows / cpy / get_rows (bit-exact) --------------------------------------------------
@pytest.mark.parametrize("dst_t,i64", [(O.F16, False), (O.F16, True), (O.F32, False)])
def test_set_rows(gpu, dst_t, i64):
    n0, rows, n = 1024, 64, 9
    src = (rng.standard_normal((n, n0)) * 10).astype(np.float32)
    src[0, :4] = [65504.0, 1e-8, -0.0, 70000.0]        # fp16 edge cases: max, underflow, signed zero, overflow -> inf
    idx = rng.permutation(rows)[:n].astype(np.int64 if i64 else np.int32)
    dst0 = rng.standard_normal((rows, n0)).astype(O.NP_OF[dst_t])
    want = dst0.copy()
    O.set_rows(O.tensor(src, O.F32, [n0, n]), O.tensor(idx, O.I64 if i64 else O.I32, [n]), O.tensor(want, dst_t, [n0, rows]))
    d = gpu.Tensor.from_numpy(dst0)
    gpu.ops.set_rows(d, gpu.Tensor.from_numpy(src), gpu.Tensor.from_numpy(idx))
    assert np.array_equal(d.numpy().view(np.uint8), want.view(np.uint8))


def test_cpy_v_cache_transposed_and_cont(gpu):
    KD, qlen, ML, n_past = 256, 5, 64, 7
    v = rng.standard_normal((qlen, KD)).astype(np.float32)
    cache0 = rng.standard_normal((KD, ML)).astype(np.float16)
    want = cache0.copy()
    O.cpy(O.tensor(v, O.F32, [qlen, KD], nb=[KD * 4, 4, KD * qlen * 4, KD * qlen * 4]),
          O.tensor(want, O.F16, [qlen, KD], nb=[2, ML * 2, ML * KD * 2, ML * KD * 2], offset=n_past * 2))
    dv = gpu.Tensor.from_numpy(v)
    dc = gpu.Tensor.from_numpy(cache0)
    gpu.ops.cpy(dv.transpose(), dc.view([qlen, KD], [2, ML * 2], offset=n_past * 2))
    assert np.array_equal(dc.numpy().view(np.uint16), want.view(np.uint16))
    # permute + cont (context layer)
    c = rng.standard_normal((3, 4, 8)).astype(np.float32)
    got = gpu.ops.cont(gpu.Tensor.from_numpy(c).permute(0, 2, 1, 3)).numpy()
    assert np.array_equal(got, np.ascontiguousarray(c.transpose(1, 0, 2)))


@pytest.mark.parametrize("t", [O.Q8_0, O.Q4_0, O.Q4_1, O.Q5_0, O.Q5_1, O.Q4_K, O.F16])
def test_weight_quantizers_on_the_device_are_byte_identical(gpu, t):
    """cllm_op_quantize_rows == the reference's from_float_ref (the loader's on-load re-quantization), byte for byte: random rows at three magnitudes plus the
    special sub-blocks (all zero, constant, all positive, ties, values on the rounding grid); a re-quantized row then dequantizes to the same floats"""
    T, L = gpu.Tensor, gpu.lib.get()
    K, rows = 4096, 37
    for scale in (1.0, 1e-3, 40.0):
        x = (rng.standard_normal((rows, K)) * scale).astype(np.float32)
        x[0, :32] = 0.0; x[0, 32:64] = 0.37; x[0, 64:96] = np.abs(x[0, 64:96]) + 0.1; x[0, 100] = -x[0, 101]
        x[1, :512] = np.round(x[1, :512] * 4) / 4
        x[2] = 0.0
        want = np.concatenate([O.quantize_ref(t, x[r]) for r in range(rows)])
        dx = T.from_numpy(x)
        out = T(t, [K, rows])
        gpu.lib.check(L.cllm_op_quantize_rows(None, t, dx.data_ptr(), out.data_ptr(), K, rows), "quantize_rows")
        got = out.numpy().view(np.uint8).reshape(-1)
        assert np.array_equal(got, want), int(np.argmax(got != want))
    assert L.cllm_op_quantize_rows(None, O.Q6_K, dx.data_ptr(), out.data_ptr(), K, rows) != 0 and b"no device quantizer" in L.cllm_last_error()


@pytest.mark.parametrize("t", [O.Q4_0, O.Q4_1, O.Q8_0, O.Q4_K, O.Q5_K, O.Q6_K, O.F16, O.F32, O.Q5_0, O.Q5_1, O.IQ4_NL, O.MXFP4, O.Q2_K, O.Q3_K, O.IQ4_XS, O.TQ1_0, O.TQ2_0, O.IQ2_XXS, O.IQ2_XS, O.IQ2_S, O.IQ3_XXS, O.IQ3_S, O.IQ1_S, O.IQ1_M])
def test_get_rows_bit_exact(gpu, t):
    n0, rows, n = 512, 30, 7
    table = rng.standard_normal((rows, n0)).astype(O.NP_OF[t]) if t in (O.F16, O.F32) else rand_blocks(t, rows, n0, rng)
    ids = rng.integers(0, rows, n).astype(np.int32)
    want = np.zeros((n, n0), np.float32)
    O.get_rows(O.tensor(table, t, [n0, rows]), O.tensor(ids, O.I32, [n]), O.tensor(want, O.F32, [n0, n]))
    got = gpu.ops.get_rows(gpu.Tensor.from_numpy(table, t, [n0, rows]), gpu.Tensor.from_numpy(ids)).numpy()
    assert np.array_equal(got.reshape(want.shape).view(np.uint32), want.view(np.uint32))


# ---- the router of a sparse-MoE block: SUM_ROWS, DIV, TOP_K (bit-exact: sums in double, IEEE division, indices) ------------------
@pytest.mark.parametrize("n0,n1,n2,k", [(8, 5, 1, 2), (8, 1, 1, 2), (64, 7, 2, 6), (3, 4, 1, 1), (160, 300, 1, 8)])
def test_moe_router_ops_bit_exact(gpu, n0, n1, n2, k):
    ops, T = gpu.ops, gpu.Tensor
    x = rng.standard_normal((n2, n1, n0)).astype(np.float32)
    x = (np.exp(x) / np.exp(x).sum(-1, keepdims=True)).astype(np.float32)
    x[0, 0, :min(n0, 3)] = x[0, 0, 0]                        # a tie: lower index first
    dx = T.from_numpy(x)
    want = np.zeros((n2, n1, 1), np.float32)
    O.sum_rows(O.tensor(x, O.F32, [n0, n1, n2]), O.tensor(want, O.F32, [1, n1, n2]))
    assert np.array_equal(ops.sum_rows(dx).numpy().reshape(want.shape).view(np.uint32), want.view(np.uint32))
    wantk = np.zeros((n2, n1, k), np.int32)
    O.top_k(O.tensor(x, O.F32, [n0, n1, n2]), O.tensor(wantk, O.I32, [k, n1, n2]))
    assert np.array_equal(ops.top_k(dx, k).numpy().reshape(wantk.shape), wantk)
    y = (np.abs(rng.standard_normal((n2, n1, 1))) + 0.1).astype(np.float32)
    wantd = np.zeros_like(x)
    O.div(O.tensor(x, O.F32, [n0, n1, n2]), O.tensor(y, O.F32, [1, n1, n2]), O.tensor(wantd, O.F32, [n0, n1, n2]))
    assert np.array_equal(ops.div(dx, T.from_numpy(y)).numpy().reshape(x.shape).view(np.uint32), wantd.view(np.uint32))


@pytest.mark.parametrize("t,K,F,E,k", [(O.Q4_K, 4096, 1024, 8, 2), (O.Q8_0, 512, 264, 4, 2), (O.Q4_0, 1024, 512, 8, 3), (O.Q4_1, 256, 64, 8, 2)])
def test_mul_mat_id_silu_mul_equals_the_four_nodes(gpu, t, K, F, E, k):
    ops, T = gpu.ops, gpu.Tensor
    wg = T.from_numpy(rand_blocks(t, F * E, K, rng), t, [K, F, E])
    wu = T.from_numpy(rand_blocks(t, F * E, K, rng), t, [K, F, E])
    x = T.from_numpy(rng.standard_normal((1, 1, K)).astype(np.float32))
    ids = T.from_numpy(rng.choice(E, k, replace=False).astype(np.int32).reshape(1, k))
    want = ops.mul(ops.mul_mat_id(wu, x, ids), ops.silu(ops.mul_mat_id(wg, x, ids))).numpy()
    got = ops.mul_mat_id_silu_mul(wg, wu, x, ids).numpy()
    assert np.array_equal(got.view(np.uint32).reshape(-1), want.view(np.uint32).reshape(-1))


@pytest.mark.parametrize("t,K,N,M", [(O.Q4_0, 4096, 512, 200), (O.Q4_K, 2048, 256, 77), (O.Q8_0, 512, 136, 64), (O.Q4_1, 1024, 264, 130), (O.Q4_K, 8192, 128, 40)])
def test_mul_mat_ex_equals_the_node_sequences(gpu, t, K, N, M):
    """cllm_op_mul_mat_ex (prefill): the norm prologue, the SiLU * up prologue / epilogue and the residual epilogue against the separate launches, bit for bit"""
    ops, T = gpu.ops, gpu.Tensor
    w = T.from_numpy(rand_blocks(t, N, K, rng), t, [K, N])
    xh = rng.standard_normal((M, K)).astype(np.float32)
    x = T.from_numpy(xh); g = T.from_numpy((1.0 + 0.1 * rng.standard_normal(K)).astype(np.float32))
    bits = lambda a: a.numpy().view(np.uint32).reshape(-1)
    # RMS_NORM -> MUL -> MUL_MAT
    want = ops.mul_mat(w, ops.rms_norm_mul(x, g, 1e-5))
    assert np.array_equal(bits(ops.mul_mat_ex(w, x, pro=1, norm_w=g, eps=1e-5)), bits(want))
    # ... -> ADD(resid), in place on the residual
    r = T.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
    want_r = ops.add(ops.mul_mat(w, x), r)
    r2 = T.from_numpy(r.numpy())
    ops.mul_mat_ex(w, x, resid=r2, dst=r2)
    assert np.array_equal(bits(r2), bits(want_r))
    # rows alternate gate_u, up_u: MUL_MAT -> (even, odd) -> SILU -> MUL in the epilogue; with the norm prologue in front
    y = ops.mul_mat(w, ops.rms_norm_mul(x, g, 1e-5))                              # [N, M]: gate at even, up at odd features
    gate = y.view([N // 2, M], [8, y.nb[1]], offset=0); up = y.view([N // 2, M], [8, y.nb[1]], offset=4)
    want_s = ops.mul(ops.silu(ops.cont(gate)), ops.cont(up))
    assert np.array_equal(bits(ops.mul_mat_ex(w, x, pro=1, norm_w=g, eps=1e-5, epi=1)), bits(want_s))
    # UNARY(SILU)(gate) -> MUL(up) -> MUL_MAT with separate gate / up tensors (pro 4), + the residual
    gt = T.from_numpy(rng.standard_normal((M, K)).astype(np.float32)); ut = T.from_numpy(rng.standard_normal((M, K)).astype(np.float32))
    want_4 = ops.add(ops.mul_mat(w, ops.mul(ops.silu(gt), ut)), r)
    assert np.array_equal(bits(ops.mul_mat_ex(w, gt, pro=4, norm_w=ut, resid=r)), bits(want_4))
    # a second projection of the same activation reuses the act rows of the previous call (pro 5)
    w2 = T.from_numpy(rand_blocks(t, N, K, rng), t, [K, N])
    ops.mul_mat_ex(w, x, pro=1, norm_w=g, eps=1e-5)
    assert np.array_equal(bits(ops.mul_mat_ex(w2, x, pro=5)), bits(ops.mul_mat(w2, ops.rms_norm_mul(x, g, 1e-5))))
    # the SiLU * up quantizer prologue (interleaved pairs in src1)
    if K * 2 <= 8192:
        x2 = T.from_numpy(rng.standard_normal((M, 2 * K)).astype(np.float32))
        ge = x2.view([K, M], [8, x2.nb[1]], offset=0); ue = x2.view([K, M], [8, x2.nb[1]], offset=4)
        want_p = ops.mul_mat(w, ops.mul(ops.silu(ops.cont(ge)), ops.cont(ue)))
        assert np.array_equal(bits(ops.mul_mat_ex(w, x2, pro=3)), bits(want_p))


@pytest.mark.parametrize("t,K,ne,k", [(O.Q4_K, 4096, 8, 2), (O.Q4_K, 256, 8, 2), (O.Q8_0, 512, 4, 2), (O.Q4_0, 1024, 16, 3), (O.Q4_1, 256, 60, 6),
                                      (O.Q4_K, 14336, 64, 8), (O.Q4_0, 4096, 7, 1), (O.Q8_0, 8192, 33, 4)])
def test_moe_router_equals_the_node_sequence(gpu, t, K, ne, k):
    """cllm_op_moe_router = RMS_NORM -> MUL -> MUL_MAT(gate) -> SOFT_MAX -> TOP_K of the reference's graph, bit for bit (device nodes and CPU oracle)"""
    ops, T = gpu.ops, gpu.Tensor
    wb = rand_blocks(t, ne, K, rng, d_scale=0.05)
    w = T.from_numpy(wb, t, [K, ne])
    xh = (rng.standard_normal(K) * 1.7).astype(np.float32); gh = (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    x = T.from_numpy(xh); g = T.from_numpy(gh)
    xn = ops.rms_norm_mul(x, g, 1e-5)
    pr = ops.soft_max(ops.mul_mat(w, xn))
    want_ids = ops.top_k(pr, k).numpy().reshape(-1)
    got_xn, got_pr, got_ids = ops.moe_router(x, g, 1e-5, w, k)
    assert np.array_equal(got_xn.numpy().view(np.uint32).reshape(-1), xn.numpy().view(np.uint32).reshape(-1))
    assert np.array_equal(got_pr.numpy().view(np.uint32).reshape(-1), pr.numpy().view(np.uint32).reshape(-1))
    assert np.array_equal(got_ids.numpy().reshape(-1), want_ids)
    # the CPU oracle on the same inputs
    n1 = np.zeros(K, np.float32); O.rms_norm(O.tensor(xh, O.F32, [K]), O.tensor(n1, O.F32, [K]), 1e-5)
    n2 = np.zeros(K, np.float32); O.mul(O.tensor(n1, O.F32, [K]), O.tensor(gh, O.F32, [K]), O.tensor(n2, O.F32, [K]))
    lg = np.zeros(ne, np.float32); O.mul_mat(O.tensor(wb, t, [K, ne]), O.tensor(n2, O.F32, [K]), O.tensor(lg, O.F32, [ne]))
    pc = np.zeros(ne, np.float32); O.soft_max(O.tensor(lg, O.F32, [ne]), None, O.tensor(pc, O.F32, [ne]))
    ic = np.zeros(k, np.int32); O.top_k(O.tensor(pc, O.F32, [ne]), O.tensor(ic, O.I32, [k]))
    assert np.array_equal(got_pr.numpy().view(np.uint32).reshape(-1), pc.view(np.uint32))
    assert np.array_equal(got_ids.numpy().reshape(-1), ic)


@pytest.mark.parametrize("t,K,F,E,k", [(O.Q4_K, 4096, 1024, 8, 2), (O.Q4_K, 256, 64, 8, 2), (O.Q8_0, 512, 264, 4, 2), (O.Q4_0, 1024, 512, 16, 3), (O.Q4_1, 256, 64, 60, 6),
                                       (O.Q4_K, 8192, 128, 33, 4), (O.Q4_K, 4096, 14336, 8, 2)])
def test_moe_router_gate_up_equals_the_two_launches(gpu, t, K, F, E, k):
    """cllm_op_moe_router_gate_up (the router redone inside the experts' gate / up launch: one launch less per sparse-MoE block) = cllm_op_moe_router followed by
    cllm_op_mul_mat_id_silu_mul on its outputs: probabilities, ids and SiLU(gate) * up of every slot, bit for bit"""
    ops, T = gpu.ops, gpu.Tensor
    wr = T.from_numpy(rand_blocks(t, E, K, rng, d_scale=0.05), t, [K, E])
    wg = T.from_numpy(rand_blocks(t, F * E, K, rng), t, [K, F, E])
    wu = T.from_numpy(rand_blocks(t, F * E, K, rng), t, [K, F, E])
    x = T.from_numpy((rng.standard_normal(K) * 1.7).astype(np.float32)); g = T.from_numpy((1.0 + 0.1 * rng.standard_normal(K)).astype(np.float32))
    xn, pr, ids = ops.moe_router(x, g, 1e-5, wr, k)
    want = ops.mul_mat_id_silu_mul(wg, wu, xn.view([K, 1, 1], [4, 4 * K, 4 * K]), ids.view([k, 1], [4, 4 * k])).numpy()
    gp, gi, gd = ops.moe_router_gate_up(x, g, 1e-5, wr, wg, wu, k)
    assert np.array_equal(gp.numpy().view(np.uint32).reshape(-1), pr.numpy().view(np.uint32).reshape(-1))
    assert np.array_equal(gi.numpy().reshape(-1), ids.numpy().reshape(-1))
    assert np.array_equal(gd.numpy().view(np.uint32).reshape(-1), want.view(np.uint32).reshape(-1))


@pytest.mark.parametrize("t,K,H,E,with_resid", [(O.Q4_K, 14336, 4096, 8, True), (O.Q4_K, 512, 256, 8, False), (O.Q8_0, 512, 264, 4, True), (O.Q4_0, 1024, 100, 8, True),
                                                (O.Q4_1, 4096, 512, 3, False), (O.Q4_K, 20480, 64, 4, True)])
def test_mul_mat_id_combine_equals_the_two_launches(gpu, t, K, H, E, with_resid):
    """cllm_op_mul_mat_id_combine = cllm_op_mul_mat_id (down experts, two slots) -> cllm_op_moe_combine, bit for bit; also in place on the residual"""
    ops, T = gpu.ops, gpu.Tensor
    w = T.from_numpy(rand_blocks(t, H * E, K, rng), t, [K, H, E])
    x = T.from_numpy(rng.standard_normal((1, 2, K)).astype(np.float32))
    pr = rng.standard_normal((1, E)).astype(np.float32)
    pr = (np.exp(pr) / np.exp(pr).sum(-1, keepdims=True)).astype(np.float32)
    p = T.from_numpy(pr)
    ids = ops.top_k(p, 2)
    rh = rng.standard_normal((1, H)).astype(np.float32)
    r = T.from_numpy(rh) if with_resid else None
    want = ops.moe_combine(ops.mul_mat_id(w, x, ids), p, ids, r).numpy()
    got = ops.mul_mat_id_combine(w, x, ids, p, r).numpy()
    assert np.array_equal(got.view(np.uint32).reshape(-1), want.view(np.uint32).reshape(-1))
    if with_resid:
        r2 = T.from_numpy(rh)
        ops.mul_mat_id_combine(w, x, ids, p, r2, dst=r2)
        assert np.array_equal(r2.numpy().view(np.uint32).reshape(-1), want.view(np.uint32).reshape(-1))


@pytest.mark.parametrize("H,k,T,ne,with_resid", [(4096, 2, 1, 8, True), (256, 2, 5, 8, False), (100, 4, 3, 16, True)])
def test_moe_combine_equals_the_node_sequence(gpu, H, k, T, ne, with_resid):
    ops, T_ = gpu.ops, gpu.Tensor
    e = T_.from_numpy(rng.standard_normal((T, k, H)).astype(np.float32))
    pr = rng.standard_normal((T, ne)).astype(np.float32)
    pr = (np.exp(pr) / np.exp(pr).sum(-1, keepdims=True)).astype(np.float32)
    p = T_.from_numpy(pr)
    ids = ops.top_k(p, k)
    r = T_.from_numpy(rng.standard_normal((T, H)).astype(np.float32))
    # node by node, as the reference's graph: GET_ROWS(probs [1, n_expert, T], ids) -> [k, T] -> SUM_ROWS -> DIV -> MUL -> ADD of slot views (-> ADD resid)
    w = ops.get_rows(p.reshape(1, ne, T), ids).reshape(k, T)
    wn = ops.div(w, ops.sum_rows(w)).reshape(1, k, T)
    y = ops.mul(e, wn)
    acc = y.view([H, T], [4, y.nb[2]], offset=0)
    for j in range(1, k):
        acc = ops.add(acc, y.view([H, T], [4, y.nb[2]], offset=j * y.nb[1]))
    if k == 1:
        acc = ops.cont(acc)
    want = (ops.add(acc, r) if with_resid else acc)
    want = ops.cont(want).numpy() if not want.is_contiguous() else want.numpy()
    got = ops.moe_combine(e, p, ids, r if with_resid else None).numpy()
    assert np.array_equal(got.view(np.uint32).reshape(-1), want.view(np.uint32).reshape(-1))


# ---- attention over strided cache views (GQA broadcast) ---------------------------------------------------
@pytest.mark.parametrize("qlen,n_past", [(1, 0), (1, 37), (1, 255), (6, 0), (5, 11), (64, 0), (200, 56), (33, 150)])
def test_attention_composite(gpu, qlen, n_past):
    hd, nh, nkv, ML = 128, 8, 2, 256
    KD, n_kv = hd * nkv, n_past + qlen
    q = rng.standard_normal((qlen, nh, hd)).astype(np.float32)
    kc = rng.standard_normal((ML, KD)).astype(np.float16)
    vc = rng.standard_normal((KD, ML)).astype(np.float16)
    sc = np.zeros((nh, qlen, n_kv), np.float32)
    ctx = np.zeros((nh, qlen, hd), np.float32)
    S = O.tensor(sc, O.F32, [n_kv, qlen, nh])
    O.mul_mat(O.tensor(kc, O.F16, [hd, n_kv, nkv], nb=[2, KD * 2, hd * 2, KD * ML * 2]),
              O.tensor(q, O.F32, [hd, qlen, nh], nb=[4, nh * hd * 4, hd * 4, nh * hd * qlen * 4]), S)
    O.scale(S, S, 1.0 / np.sqrt(hd))
    O.diag_mask_inf(S, S, n_past)
    O.soft_max(S, None, S)
    O.mul_mat(O.tensor(vc, O.F16, [n_kv, hd, nkv], nb=[2, ML * 2, ML * hd * 2, ML * KD * 2]), S, O.tensor(ctx, O.F32, [hd, qlen, nh]))
    want = np.ascontiguousarray(ctx.transpose(1, 0, 2)).reshape(qlen, nh * hd)

    ops = gpu.ops
    dq, dk, dv = gpu.Tensor.from_numpy(q), gpu.Tensor.from_numpy(kc), gpu.Tensor.from_numpy(vc)
    Kv = dk.view([hd, n_kv, nkv], [2, KD * 2, hd * 2])
    Qv = dq.permute(0, 2, 1, 3)
    s = ops.mul_mat(Kv, Qv)
    p = ops.scale_mask_soft_max(s, 1.0 / np.sqrt(hd), n_past)
    Vv = dv.view([n_kv, hd, nkv], [2, ML * 2, ML * hd * 2])
    c = ops.mul_mat(Vv, p)
    got = ops.cont(c.permute(0, 2, 1, 3)).numpy().reshape(qlen, nh * hd)
    # the probabilities are rounded to fp16 before V.P (CPU semantics): a 1e-7 difference in P can flip one of those
    # roundings (5e-4 relative on that element), hence the looser bound on the composite
    assert rel_err(got, want) < 3e-4


# ---- the single-token attention block in one call == the node sequence chatllm emits (KVCacheAttention, src/layers.cpp:3044-3123) --------
@pytest.mark.parametrize("hd,nh,nkv,ML,n_past,mode,table", [
    (128, 32, 8, 1024, 0, 0, True), (128, 32, 8, 1024, 37, 0, True), (128, 32, 8, 1024, 300, 2, True), (128, 32, 8, 1024, 511, 0, True),
    (128, 8, 8, 2048, 700, 0, True), (128, 8, 8, 2048, 1023, 0, True),       # n_kv 1024: the last context the one-launch kernel takes; R2 = 1
    (128, 8, 8, 2048, 1500, 0, True),       # above the long-context threshold: three launches, R2 = 1
    (128, 32, 8, 4096, 3000, 2, True),      # long, GQA 4
    (128, 32, 8, 8192, 8190, 0, True),      # 255 whole chunks + 31 leftovers, the last slice of positions ragged
    (128, 16, 2, 2048, 1055, 2, True),      # GQA 8
    (64, 8, 4, 4096, 2049, 0, True),        # head size 64, two leftovers
    (128, 32, 8, 16384, 12345, 0, True),
    (64, 4, 2, 64, 11, 0, True), (64, 4, 2, 64, 63, 2, True),
    (128, 32, 8, 1024, 100, 0, False),      # no table: the general kernel computes cos/sin itself
    (96, 6, 2, 256, 40, 2, True)])          # head size the compact kernels do not take -> general kernel
def test_rope_kv_attn_decode_equals_the_node_sequence(gpu, hd, nh, nkv, ML, n_past, mode, table):
    ops, T = gpu.ops, gpu.Tensor
    QD, KD, n_kv, fb = hd * nh, hd * nkv, n_past + 1, 500000.0
    qkv = rng.standard_normal(QD + 2 * KD).astype(np.float32)
    kc0 = rng.standard_normal((ML, KD)).astype(np.float16)
    vc0 = rng.standard_normal((KD, ML)).astype(np.float16)
    pos = T.from_numpy(np.array([n_past], np.int32))

    # -- node by node: ROPE(k) -> SET_ROWS, TRANSPOSE(v) -> CPY, ROPE(q), MUL_MAT(K,Q), SCALE+MASK+SOFT_MAX, MUL_MAT(V,P), PERMUTE+CONT
    dk, dv = T.from_numpy(kc0), T.from_numpy(vc0)
    q = T.from_numpy(qkv[:QD].reshape(1, nh, hd))
    k = T.from_numpy(qkv[QD:QD + KD].reshape(1, nkv, hd))
    v = T.from_numpy(qkv[QD + KD:].reshape(1, KD))
    ops.cpy(v.transpose(), dv.view([1, KD], [2, ML * 2], offset=n_past * 2))
    kr = ops.rope_ext(k, pos, None, hd, mode, freq_base=fb, inplace=True)
    ops.set_rows(dk.view([KD, ML], [2, KD * 2]), kr.reshape(KD, 1), pos)
    qr = ops.rope_ext(q, pos, None, hd, mode, freq_base=fb, inplace=True)
    s = ops.mul_mat(dk.view([hd, n_kv, nkv], [2, KD * 2, hd * 2]), qr.permute(0, 2, 1, 3))
    p = ops.scale_mask_soft_max(s, float(np.float32(1.0) / np.sqrt(np.float32(hd))), n_past)
    c = ops.mul_mat(dv.view([n_kv, hd, nkv], [2, ML * 2, ML * hd * 2]), p)
    want = ops.cont(c.permute(0, 2, 1, 3)).numpy().reshape(QD)
    want_k, want_v = dk.numpy().view(np.uint16), dv.numpy().view(np.uint16)

    # -- post-RoPE form (caches already written): cllm_op_attn_decode
    if ML <= 4096:
        got1 = ops.attn_decode(qr, pos, nh, nkv, hd, dk, dv, ML).numpy().reshape(QD)
        assert np.array_equal(got1.view(np.uint32), want.view(np.uint32))

    # -- everything in one call, on fresh caches
    fk, fv = T.from_numpy(kc0), T.from_numpy(vc0)
    got = ops.rope_kv_attn_decode(T.from_numpy(qkv), pos, n_kv, nh, nkv, hd, mode, fb, fk, fv, ML, table=table).numpy().reshape(QD)
    assert np.array_equal(fk.numpy().view(np.uint16), want_k)
    assert np.array_equal(fv.numpy().view(np.uint16), want_v)
    # every context length: the one-launch kernel up to 1024 cached positions, the three split launches of attn_long.hip beyond -- both in ggml_vec_dot_f16's order
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("hd,nh,nkv,n_past", [(128, 32, 8, 299), (128, 32, 8, 318), (64, 8, 4, 40), (128, 16, 2, 480)])
def test_attn_decode_leftover_sums_over_a_wide_exponent_range(gpu, hd, nh, nkv, n_past):
    """ggml_vec_dot_f16 adds the n mod 32 leftover products one by one in double.  k_attn_dec proves per output row that no addition can round (exponents of the terms
    within 23 binades) and then adds them as a tree, else keeps the serial loop.  Here the leftover positions of every second V^T row hold fp16 SUBNORMALS (products ~2^-28
    beside a reduced sum ~2^-4: the serial path), the other rows ordinary values (the tree) -- each against the node sequence, to the bit"""
    ops, T = gpu.ops, gpu.Tensor
    ML, mode, fb = 1024, 0, 500000.0
    QD, KD, n_kv = hd * nh, hd * nkv, n_past + 1
    r = np.random.default_rng(n_past)
    qkv = r.standard_normal(QD + 2 * KD).astype(np.float32)
    kc0 = r.standard_normal((ML, KD)).astype(np.float16)
    vc0 = r.standard_normal((KD, ML)).astype(np.float16)
    npos = n_kv & ~31
    tiny = (r.integers(-15, 16, (KD // 2, n_kv - npos)) * 2.0 ** -24).astype(np.float16)       # multiples of the smallest fp16 subnormal
    vc0[0::2, npos:n_kv] = tiny
    qkv[QD + KD:][0::2] *= 2.0 ** -22                                                              # the new token's v (the last leftover) as well
    pos = T.from_numpy(np.array([n_past], np.int32))
    dk, dv = T.from_numpy(kc0), T.from_numpy(vc0)
    q = T.from_numpy(qkv[:QD].reshape(1, nh, hd))
    k = T.from_numpy(qkv[QD:QD + KD].reshape(1, nkv, hd))
    v = T.from_numpy(qkv[QD + KD:].reshape(1, KD))
    ops.cpy(v.transpose(), dv.view([1, KD], [2, ML * 2], offset=n_past * 2))
    kr = ops.rope_ext(k, pos, None, hd, mode, freq_base=fb, inplace=True)
    ops.set_rows(dk.view([KD, ML], [2, KD * 2]), kr.reshape(KD, 1), pos)
    qr = ops.rope_ext(q, pos, None, hd, mode, freq_base=fb, inplace=True)
    s = ops.mul_mat(dk.view([hd, n_kv, nkv], [2, KD * 2, hd * 2]), qr.permute(0, 2, 1, 3))
    p = ops.scale_mask_soft_max(s, float(np.float32(1.0) / np.sqrt(np.float32(hd))), n_past)
    c = ops.mul_mat(dv.view([n_kv, hd, nkv], [2, ML * 2, ML * hd * 2]), p)
    want = ops.cont(c.permute(0, 2, 1, 3)).numpy().reshape(QD)
    fk, fv = T.from_numpy(kc0), T.from_numpy(vc0)
    got = ops.rope_kv_attn_decode(T.from_numpy(qkv), pos, n_kv, nh, nkv, hd, mode, fb, fk, fv, ML, table=True).numpy().reshape(QD)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int(np.sum(got.view(np.uint32) != want.view(np.uint32)))
    assert np.array_equal(fv.numpy().view(np.uint16), dv.numpy().view(np.uint16))


def test_rope_kv_attn_decode_rejects_what_it_cannot_do(gpu):
    L, T = gpu.lib.get(), gpu.Tensor
    assert L.cllm_attn_decode_supported(32, 8, 128, 1024) == 1
    assert L.cllm_attn_decode_supported(32, 8, 128, 1 << 17) == 0          # scores of one head no longer fit the LDS
    assert L.cllm_attn_decode_supported(32, 5, 128, 1024) == 0
    assert L.cllm_attn_decode_wsize(100, 32, 2048) == 0 and L.cllm_attn_decode_wsize(512, 32, 2048) == 0 and L.cllm_attn_decode_wsize(513, 32, 2048) == 32 * 2048 * 6   # (CLLM_ATTN_LONG, default 512)
    x = T.from_numpy(np.zeros(4096, np.float32)); pos = T.from_numpy(np.zeros(1, np.int32)); kc = T.from_numpy(np.zeros((64, 256), np.float16))
    rc = L.cllm_op_rope_kv_attn_decode(None, x.data_ptr(), pos.data_ptr(), None, 1e4, 1, 4, 2, 128, 1, kc.data_ptr(), kc.data_ptr(), 64, x.data_ptr(), None, 0)
    assert rc != 0 and b"rope mode" in L.cllm_last_error()
    rc = L.cllm_op_rope_kv_attn_decode(None, x.data_ptr(), pos.data_ptr(), None, 1e4, 65, 4, 2, 128, 0, kc.data_ptr(), kc.data_ptr(), 64, x.data_ptr(), None, 0)
    assert rc != 0


# ---- fused single-token patterns (cllm_op_mul_mat_vec_fused) == the node sequence they replace, to the bit ----------------------
@pytest.mark.parametrize("t,K,N,pro", [(O.Q4_K, 4096, 512, 1), (O.Q4_K, 8192, 256, 1), (O.Q4_K, 14336, 256, 4), (O.Q4_K, 29440, 128, 4),
                                       (O.Q8_0, 29568, 128, 4), (O.Q4_0, 4096, 384, 1), (O.Q8_0, 2048, 100, 2), (O.Q4_K, 256, 33, 4),
                                       (O.Q4_1, 4096, 200, 1), (O.Q4_1, 14336, 64, 4), (O.Q4_1, 1024, 33, 2),
                                       (O.Q4_K, 4096, 512, 2), (O.Q4_K, 14336, 256, 2), (O.Q4_K, 20480, 64, 2), (O.Q4_K, 256, 33, 2), (O.Q4_K, 1280, 40, 2)])
def test_mul_mat_vec_fused_equals_the_node_sequence(gpu, t, K, N, pro):
    ops, T = gpu.ops, gpu.Tensor
    w = T.from_numpy(rand_blocks(t, N, K, rng), t, [K, N])
    x = T.from_numpy(rng.standard_normal((1, K)).astype(np.float32))
    g = T.from_numpy((1 + 0.1 * rng.standard_normal((1, K))).astype(np.float32))
    r = T.from_numpy(rng.standard_normal((1, N)).astype(np.float32))
    if pro == 1:
        act = ops.rms_norm_mul(x, T.from_numpy(g.numpy().reshape(K)), 1e-5)
    elif pro == 4:
        act = ops.silu_mul(x, g)
    else:
        act = x
    want = ops.add(ops.mul_mat(w, act), r).numpy()
    out = T(gpu.F32, [N, 1])
    cw = w.c()
    gpu.lib.check(gpu.lib.get().cllm_op_mul_mat_vec_fused(None, C.byref(cw), pro, x.data_ptr(), g.data_ptr() if pro != 2 else None, 1e-5, 0,
                                                           r.data_ptr(), out.data_ptr()), "fused")
    assert np.array_equal(out.numpy(), want)


@pytest.mark.parametrize("K", [1280, 14336, 17408])
def test_fused_plain_quantize_prologue_on_ties_and_zero_blocks(gpu, K):
    """the 16-values-per-lane quantize_row_q8_K of the decode mat-vec's plain-quantize prologue (quant16_q8_K): super-blocks whose largest magnitude occurs with BOTH
    signs (the reference keeps the sign of the FIRST occurrence, ggml-quants.c:2562-2566) -- in different lanes, inside one lane, negative first, positive first --
    an all-zero super-block, a super-block of one value; against quantize (oracle order) + mat-vec, to the bit"""
    ops, T = gpu.ops, gpu.Tensor
    t, N = O.Q4_K, 96
    w = T.from_numpy(rand_blocks(t, N, K, rng), t, [K, N])
    x = rng.standard_normal(K).astype(np.float32)
    m = np.float32(7.25)
    x[3], x[200] = -m, m                      # block 0: negative first, other lane
    x[256 + 5], x[256 + 130] = m, -m          # block 1: positive first
    x[512:768] = 0.0                          # block 2: all zero
    x[768 + 34], x[768 + 41] = -m, m          # block 3: both inside one lane's 16 values, negative first
    x[1024 + 41], x[1024 + 34] = -m, m        # block 4: inside one lane, positive first
    nb = K // 256
    x[256 * (nb - 1): 256 * nb] = np.float32(-0.375)      # last block: one value everywhere (every element is "the" maximum; the first one counts)
    if nb > 8:
        x[256 * 7: 256 * 8] = np.float32(0.0); x[256 * 7 + 255] = np.float32(-3.0)        # a single non-zero element at the block's end
    xt = T.from_numpy(x.reshape(1, K))
    r = T.from_numpy(rng.standard_normal((1, N)).astype(np.float32))
    want = ops.add(ops.mul_mat(w, xt), r).numpy()
    out = T(gpu.F32, [N, 1])
    cw = w.c()
    gpu.lib.check(gpu.lib.get().cllm_op_mul_mat_vec_fused(None, C.byref(cw), 2, xt.data_ptr(), None, 1e-5, 0, r.data_ptr(), out.data_ptr()), "fused")
    assert np.array_equal(out.numpy(), want)
    # and against the oracle's quantize_row_q8_K + vec_dot (AVX2 order) directly
    wb = rand_blocks(t, N, K, np.random.default_rng(77))
    w2 = T.from_numpy(wb, t, [K, N])
    ref = np.zeros(N, np.float32)
    O.mul_mat(O.tensor(wb, t, [K, N]), O.tensor(x, O.F32, [K, 1]), O.tensor(ref, O.F32, [N, 1]))
    cw2 = w2.c()
    gpu.lib.check(gpu.lib.get().cllm_op_mul_mat_vec_fused(None, C.byref(cw2), 2, xt.data_ptr(), None, 1e-5, 0, None, out.data_ptr()), "fused")
    assert np.array_equal(out.numpy().reshape(-1).view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("t,K,F", [(O.Q4_K, 4096, 14336), (O.Q4_0, 4096, 512), (O.Q8_0, 1024, 264), (O.Q4_K, 256, 512), (O.Q4_1, 2048, 1024)])
def test_packed_rows_merge_mat_vecs_without_changing_a_bit(gpu, t, K, F):
    """cllm_pack_rows + one launch == the separate launches: q|k|v concatenated; gate/up interleaved with the SiLU*up epilogue"""
    ops, T, L = gpu.ops, gpu.Tensor, gpu.lib.get()
    rb = O.row_size(t, K) if hasattr(O, "row_size") else L.cllm_row_size(t, K)
    x = T.from_numpy(rng.standard_normal((1, K)).astype(np.float32))
    g = T.from_numpy((1 + 0.1 * rng.standard_normal(K)).astype(np.float32))
    act = ops.rms_norm_mul(x, g, 1e-5)

    def fused(w, nrows, epi, n_out):
        out = T(gpu.F32, [n_out, 1])
        cw = T(t, [K, nrows], buf=w.buf).c()
        gpu.lib.check(L.cllm_op_mul_mat_vec_fused(None, C.byref(cw), 1, x.data_ptr(), g.data_ptr(), 1e-5, epi, None, out.data_ptr()), "fused")
        return out.numpy().reshape(n_out)

    def pack(ws, interleave):
        rows = [w.ne[1] for w in ws]
        dst = T(t, [K, sum(rows)])
        srcs = (C.c_void_p * len(ws))(*[w.data_ptr().value for w in ws])
        nr = (C.c_int64 * len(ws))(*rows)
        gpu.lib.check(L.cllm_pack_rows(None, dst.data_ptr(), srcs, nr, len(ws), rb, interleave), "pack")
        return dst

    # q | k | v
    ws = [T.from_numpy(rand_blocks(t, n, K, rng), t, [K, n]) for n in (F // 2, 64, 64)]
    want = np.concatenate([ops.mul_mat(w, act).numpy().reshape(-1) for w in ws])
    assert np.array_equal(fused(pack(ws, 0), F // 2 + 128, 0, F // 2 + 128).view(np.uint32), want.view(np.uint32))
    # gate / up -> silu(gate) * up
    wg, wu = (T.from_numpy(rand_blocks(t, F, K, rng), t, [K, F]) for _ in range(2))
    want = ops.mul(ops.silu(ops.mul_mat(wg, act)), ops.mul_mat(wu, act)).numpy().reshape(-1)
    assert np.array_equal(fused(pack([wg, wu], 1), 2 * F, 1, F).view(np.uint32), want.view(np.uint32))


# ---- randomized differential run (tools/fuzz_parity.py): random shapes / types / column counts / context lengths, bit equality with the oracle ----
def test_fuzz_parity_thirty_seconds(gpu):
    """the fuzzer behind profiles/r03_fuzz_parity.txt as a test: 30 s of seeded random cases (MUL_MAT of every weight type and column count, the fused decode
    mat-vec forms, MUL_MAT_ID, the single-token attention block, the device weight quantizers) -- every result must have the oracle's bits"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--seconds", "30", "--seed", "20260924"], capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert "0 mismatches" in r.stdout or "mismatches: 0" in r.stdout or "0 mismatch" in r.stdout, tail
