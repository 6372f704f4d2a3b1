"""Committed golden vectors produced by the REAL reference (tests/golden/make_golden.py: ggml CPU backend for the ops,
the chatllm.cpp host `ref_chat` for whole models).  CPU part: the oracle reproduces them (this is what pins the oracle on
machines without /root/reference).  GPU part (-m gpu): the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

import oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ops_reference.npz"))
M = np.load(os.path.join(os.path.dirname(__file__), "golden", "tiny_llama3_reference.npz"))
G41 = np.load(os.path.join(os.path.dirname(__file__), "golden", "q4_1_reference.npz"))       # Q4_1 vectors (added later: own file)
TYPES = ((O.Q4_K, "q4_K"), (O.Q4_0, "q4_0"), (O.Q8_0, "q8_0"), (O.Q4_1, "q4_1"))


def _g(key):
    return G41[key] if key in G41.files else G[key]


def _model(name):
    if name == "q4_1":
        return G41["model_prompt"], G41["model_ids"], G41["model_logits"]
    return M["prompt"], M[f"{name}_ids"], M[f"{name}_logits"]


def rel(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


# ---------------------------------------------------------------- CPU: oracle vs golden
CV = np.load(os.path.join(os.path.dirname(__file__), "golden", "convert_py_reference.npz"))      # bytes written by the reference's converter (tests/golden/make_convert_golden.py)


@pytest.mark.parametrize("t,name", [(O.Q8_0, "q8_0"), (O.Q4_0, "q4_0"), (O.Q4_1, "q4_1"), (O.Q4_K, "q4_k")])
def test_convert_py_quantizers(t, name):
    """the reference's torch quantizers (convert.py:328-568 -- what writes every stock chatllm model file; quantize_q4_k claims "byte-identical to the original" at
    :491-494) against the oracle's from_float_ref restatement (itself pinned byte for byte to libggml-base.so): a second, independent statement of the encodings.
    Q8_0 / Q4_1: every byte equal.  Q4_0: the converter divides by the scale and rounds half to even (`(x / scale + 8).round()`, convert.py:346) where ggml multiplies
    by 1 / d and truncates x + 8.5 (ggml-quants.c:36-70), and it has no zero-block guard -- ties and all-zero blocks differ, by at most one step.  Q4_K: the
    converter's search is vectorised float32 torch, the C reference's a scalar loop.  For those two the differing rows are counted and must decode to the same
    weights within two quantization steps (so a stock model file and an on-load re-quantization of the same floats are the same model, not the same bytes)."""
    n_rows = n_same = 0
    exact = t in (O.Q8_0, O.Q4_1)
    for key in [k for k in CV.files if k.startswith("x_")]:
        x, want = CV[key], CV[name + key[1:]]
        for r in range(x.shape[0]):
            got = O.quantize_ref(t, x[r])
            n_rows += 1
            if np.array_equal(got, want[r]):
                n_same += 1
                continue
            assert not exact, (name, key, r, int(np.argmax(got != want[r])))
            K = x.shape[1]
            a, b = O.dequantize(t, got, K), O.dequantize(t, want[r], K)
            amax32 = np.abs(x[r]).reshape(-1, 32).max(axis=1).repeat(32)
            step = amax32 / (8.0 if t == O.Q4_0 else 15.0) + 1e-12
            if t == O.Q4_K:                          # sub-block scales are 6-bit fractions of the super-block's largest: a step of that grid on top
                step = step + np.abs(x[r]).reshape(-1, 256).max(axis=1).repeat(256) / 63.0
            assert np.all(np.abs(a - b) <= 2.0 * step + 1e-6), (key, r, float(np.max(np.abs(a - b) / step)))
    if exact:
        assert n_same == n_rows
    print(f"{name}: {n_same} of {n_rows} rows byte-identical to convert.py")


@pytest.mark.parametrize("K", [256, 4096])
def test_oracle_quantizers_match_golden_bytes(K):
    x = G[f"quant_x_{K}"]
    assert np.array_equal(O.quantize_q8_0(x), G[f"quant_q8_0_{K}"])
    assert np.array_equal(O.quantize_q8_K(x), G[f"quant_q8_K_{K}"])
    assert np.array_equal(O.quantize_q8_1(G41[f"quant_x_{K}"]), G41[f"quant_q8_1_{K}"])


@pytest.mark.parametrize("t,name", TYPES)
def test_oracle_mul_mat_and_dequant_match_golden(t, name):
    assert np.array_equal(O.dequantize(t, _g(f"mm_{name}_1_w")[0], 1024), _g(f"dequant_{name}"))
    for Mc in (1, 12):
        w, x, y = _g(f"mm_{name}_{Mc}_w"), _g(f"mm_{name}_{Mc}_x"), _g(f"mm_{name}_{Mc}_y")
        got = np.zeros_like(y)
        O.mul_mat(O.tensor(w, t, [1024, 48]), O.tensor(x, O.F32, [1024, Mc]), O.tensor(got, O.F32, [48, Mc]))
        assert rel(got, y) < 1e-5


def test_oracle_norm_rope_softmax_silu_attention_match_golden():
    got = np.zeros_like(G["rms_y"])
    O.rms_norm(O.tensor(G["rms_x"], O.F32, [512, 3]), O.tensor(got, O.F32, [512, 3]), 1e-5)
    assert np.array_equal(got, G["rms_y"])
    got = np.zeros_like(G["softmax_y"])
    O.soft_max(O.tensor(G["softmax_x"], O.F32, [77, 4]), None, O.tensor(got, O.F32, [77, 4]))
    assert np.allclose(got, G["softmax_y"], rtol=2e-7, atol=0)
    got = np.zeros_like(G["silu_y"])
    O.silu(O.tensor(G["silu_x"], O.F32, [77, 4]), O.tensor(got, O.F32, [77, 4]))
    assert np.array_equal(got[:, :72], G["silu_y"][:, :72]) and np.allclose(got, G["silu_y"], rtol=1e-6, atol=0)
    for mode in (0, 2):
        got = np.zeros_like(G["rope_x"])
        O.rope(O.tensor(G["rope_x"], O.F32, [128, 3, 5]), G["rope_pos"], None, O.tensor(got, O.F32, [128, 3, 5]), 128, mode, 500000.0)
        assert np.allclose(got, G[f"rope_y_mode{mode}"], rtol=3e-7, atol=3e-7)


@pytest.mark.parametrize("t,name", ((O.Q4_K, "q4_k"), (O.Q4_0, "q4_0"), (O.Q8_0, "q8_0"), (O.Q4_1, "q4_1")))
def test_oracle_whole_model_matches_the_reference_host(pkg, t, name):
    """the oracle's whole-model walk (AVX2-order dot products, the reference build's RoPE contraction) vs chatllm.cpp itself on the same GGMM
    file (committed logits of its CPU run): BIT-IDENTICAL, prompt chunk and every decode step"""
    cfg = pkg.synth.config("tiny", max_len=64)
    m = O.Llama(cfg, pkg.synth.make_model(cfg, t, seed=1234))
    prompt, ids, logits = _model(name)
    lg = m.forward(prompt)
    for s in range(13):
        assert np.array_equal(lg.view(np.uint32), logits[s].view(np.uint32)), (s, float(np.max(np.abs(lg - logits[s]))))
        assert int(np.argmax(lg)) == int(ids[s])
        if s < 12:
            lg = m.forward([int(ids[s])])


# ---------------------------------------------------------------- GPU: HIP path vs golden
@pytest.mark.gpu
@pytest.mark.parametrize("K", [256, 4096])
def test_gpu_quantizers_match_golden_bytes(gpu, K):
    x = gpu.Tensor.from_numpy(G[f"quant_x_{K}"])
    y0, yk = gpu.Tensor(gpu.I32, [K // 32 * 34 // 4 + 8]), gpu.Tensor(gpu.I32, [K // 256 * 292 // 4 + 8])
    gpu.lib.check(gpu.lib.get().cllm_quantize_row_q8_0(None, x.data_ptr(), y0.data_ptr(), K), "q8_0")
    gpu.lib.check(gpu.lib.get().cllm_quantize_row_q8_K(None, x.data_ptr(), yk.data_ptr(), K), "q8_K")
    assert np.array_equal(y0.raw()[: K // 32 * 34], G[f"quant_q8_0_{K}"])
    assert np.array_equal(yk.raw()[: K // 256 * 292], G[f"quant_q8_K_{K}"])
    x1, y1 = gpu.Tensor.from_numpy(G41[f"quant_x_{K}"]), gpu.Tensor(gpu.I32, [K // 32 * 36 // 4 + 8])
    gpu.lib.check(gpu.lib.get().cllm_quantize_row_q8_1(None, x1.data_ptr(), y1.data_ptr(), K), "q8_1")
    assert np.array_equal(y1.raw()[: K // 32 * 36], G41[f"quant_q8_1_{K}"])


@pytest.mark.gpu
@pytest.mark.parametrize("t,name", TYPES)
def test_gpu_mul_mat_matches_golden(gpu, t, name):
    for Mc in (1, 12):                                  # GEMV kernel and int8-MFMA GEMM kernel
        w, x, y = _g(f"mm_{name}_{Mc}_w"), _g(f"mm_{name}_{Mc}_x"), _g(f"mm_{name}_{Mc}_y")
        got = gpu.ops.mul_mat(gpu.Tensor.from_numpy(w, t, [1024, 48]), gpu.Tensor.from_numpy(x)).numpy().reshape(y.shape)
        assert rel(got, y) < 1e-5


@pytest.mark.gpu
def test_gpu_attention_composite_matches_golden(gpu):
    hd, nh, nkv, ML, qlen, n_past = [int(v) for v in G["attn_dims"]]
    KD, n_kv = hd * nkv, n_past + qlen
    ops = gpu.ops
    dq, dk, dv = gpu.Tensor.from_numpy(G["attn_q"]), gpu.Tensor.from_numpy(G["attn_kc"]), gpu.Tensor.from_numpy(G["attn_vc"])
    s = ops.mul_mat(dk.view([hd, n_kv, nkv], [2, KD * 2, hd * 2]), dq.permute(0, 2, 1, 3))
    p = ops.scale_mask_soft_max(s, 1.0 / np.sqrt(hd), n_past)
    c = ops.mul_mat(dv.view([n_kv, hd, nkv], [2, ML * 2, ML * hd * 2]), p)
    got = ops.cont(c.permute(0, 2, 1, 3)).numpy().reshape(qlen, nh * hd)
    assert rel(got, G["attn_out"]) < 3e-4               # fp16 rounding of P can flip (see test_gpu_ops)


@pytest.mark.gpu
@pytest.mark.parametrize("t,name", ((O.Q4_K, "q4_k"), (O.Q4_0, "q4_0"), (O.Q8_0, "q8_0"), (O.Q4_1, "q4_1")))
def test_gpu_whole_model_against_the_reference_host(gpu, t, name):
    """decoder runner vs chatllm.cpp's own CPU run on the same synthetic GGMM weights (committed logits): BIT-IDENTICAL -- the prompt chunk
    (9 tokens: the exact-order multi-column kernels), the node-by-node decode path and the fused decode path; greedy ids follow"""
    cfg = gpu.synth.config("tiny", max_len=64)
    prompt, ids, logits = _model(name)
    for fused in (False, True):
        m = gpu.Llama(cfg, gpu.synth.make_model(cfg, t, seed=1234))
        lg = m.forward(prompt)
        for s in range(13):
            assert np.array_equal(lg.view(np.uint32), logits[s].view(np.uint32)), (fused, s, float(np.max(np.abs(lg - logits[s]))))
            assert int(np.argmax(lg)) == int(ids[s])
            if s < 12:
                lg = m.decode_fused_logits(int(ids[s])) if fused else m.forward([int(ids[s])])
        m.close()
