"""Model-level parity on the GPU: the C++ decoder runner (csrc/decoder.hip through the C ABI) against the CPU
oracle's whole-model restatement on identical synthetic quantized weights.
Bar (BASELINE.json north_star): greedy token ids identical, logits within 1e-3."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3


def run_pair(gpu, cfg, wtype, prompt, n_decode):
    w = gpu.synth.make_model(cfg, wtype, seed=1234)
    ref = O.Llama(cfg, w)
    dev = gpu.Llama(cfg, w)
    lr = ref.forward(prompt)
    lg = dev.forward(prompt)
    diffs = [float(np.max(np.abs(lr - lg)))]
    margins = []
    ids_r, ids_g = [], []
    for _ in range(n_decode):
        tr, tg = int(np.argmax(lr)), int(np.argmax(lg))
        top2 = np.partition(lr, -2)[-2:]
        margins.append(float(top2[1] - top2[0]))
        ids_r.append(tr)
        ids_g.append(tg)
        lr = ref.forward([tr])
        lg = dev.forward([tr])          # teacher-forced on the reference ids so one near-tie cannot cascade
        diffs.append(float(np.max(np.abs(lr - lg))))
    dev.close()
    return ids_r, ids_g, diffs, margins


@pytest.mark.parametrize("wtype", [O.Q8_0, O.Q4_0, O.Q4_K])
def test_tiny_llama3_prefill_and_decode(gpu, wtype):
    cfg = gpu.synth.config("tiny", max_len=64)
    prompt = np.random.default_rng(5).integers(0, cfg["vocab"], 9).astype(np.int32)
    ids_r, ids_g, diffs, margins = run_pair(gpu, cfg, wtype, prompt, 24)
    assert max(diffs) < LOGIT_TOL, diffs
    # greedy ids identical wherever the reference's own top-1 margin exceeds the logit tolerance
    for a, b, m in zip(ids_r, ids_g, margins):
        assert a == b or m < 2 * LOGIT_TOL
    assert ids_r == ids_g


def test_qwen2_style_neox_bias_mixed_quant(gpu):
    """Qwen2 features: NEOX rope, qkv bias, and a down_proj that falls back to Q8_0 because ffn % 256 != 0"""
    cfg = gpu.synth.config("tiny", max_len=48, rope_mode=2, qkv_bias=1, rope_theta=1e6, ffn=544)   # 544 % 256 != 0, % 32 == 0
    prompt = np.random.default_rng(6).integers(0, cfg["vocab"], 5).astype(np.int32)
    ids_r, ids_g, diffs, _ = run_pair(gpu, cfg, O.Q4_K, prompt, 12)
    assert max(diffs) < LOGIT_TOL, diffs
    assert ids_r == ids_g


def test_decode_greedy_loop_matches_stepwise(gpu):
    cfg = gpu.synth.config("tiny", max_len=64)
    w = gpu.synth.make_model(cfg, O.Q4_K)
    prompt = np.array([1, 5, 9, 200, 17], np.int32)
    a = gpu.Llama(cfg, w)
    first = int(np.argmax(a.forward(prompt)))
    toks = [first]
    for _ in range(15):
        toks.append(int(np.argmax(a.forward([toks[-1]]))))
    b = gpu.Llama(cfg, w)
    b.forward(prompt)
    got = b.decode_greedy(first, 15)
    assert list(got) == toks[1:]
    a.close()
    b.close()


def test_small_model_gqa_long_context(gpu):
    """4 layers, head_dim 128, GQA 4:1, 40-token prompt then decode: exercises the 128-wide attention rows and ragged n_kv"""
    cfg = gpu.synth.config("small", max_len=96)
    prompt = np.random.default_rng(8).integers(0, cfg["vocab"], 40).astype(np.int32)
    ids_r, ids_g, diffs, _ = run_pair(gpu, cfg, O.Q4_K, prompt, 6)
    assert max(diffs) < LOGIT_TOL, diffs
    assert ids_r == ids_g


def test_context_overflow_is_rejected(gpu):
    cfg = gpu.synth.config("tiny", max_len=8)
    m = gpu.Llama(cfg, gpu.synth.make_model(cfg, O.Q8_0))
    with pytest.raises(gpu.lib.CllmError):
        m.forward(np.zeros(9, np.int32))
    with pytest.raises(gpu.lib.CllmError):
        m.forward(np.array([cfg["vocab"]], np.int32))     # token id out of range
    m.close()
