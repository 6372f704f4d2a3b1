"""CPU-only: host-side logic -- synthetic block writers, byte counts, tensor metadata views."""
import numpy as np

import oracle as O


def test_synth_q4_K_scale_packing_round_trips(pkg):
    rng = np.random.default_rng(3)
    w = pkg.synth.quant_blocks(O.Q4_K, 4, 1024, rng, 0.02)
    assert w.shape == (4, 4 * 144)
    x = O.dequantize(O.Q4_K, w[0], 1024)
    assert np.isfinite(x).all() and 0.005 < x.std() < 0.08 and abs(x.mean()) < 0.01


def test_synth_q8_0_q4_0_statistics(pkg):
    rng = np.random.default_rng(4)
    for t in (O.Q8_0, O.Q4_0, O.Q4_1):
        w = pkg.synth.quant_blocks(t, 2, 2048, rng, 0.05)
        x = O.dequantize(t, w[1], 2048)
        assert 0.02 < x.std() < 0.1 and abs(x.mean()) < 0.02


def test_synth_is_deterministic_per_tensor(pkg):
    a = pkg.synth.make_tensor("layers.3.wq", O.Q4_K, 8, 512)
    b = pkg.synth.make_tensor("layers.3.wq", O.Q4_K, 8, 512)
    c = pkg.synth.make_tensor("layers.4.wq", O.Q4_K, 8, 512)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_bytes_per_token_match_survey(pkg):
    cfg = pkg.synth.config("llama3-8b")
    w = pkg.synth.weight_bytes_per_token(cfg, O.Q4_K)
    assert w - (2 * 32 + 1) * 4096 * 4 == 4_221_370_368
    assert pkg.synth.weight_bytes_per_token(cfg, O.Q8_0) - (2 * 32 + 1) * 4096 * 4 == 7_973_699_584
    assert pkg.synth.kv_bytes_per_token(cfg, 0) == 131_072
    q = pkg.synth.config("qwen2-72b")
    assert pkg.synth.down_type(q, O.Q4_K) == O.Q8_0
    assert pkg.synth.weight_bytes_per_token(q, O.Q4_K) - (2 * 80 + 1) * 8192 * 4 == 49_884_168_192


def test_oracle_llama_is_causal_and_deterministic(pkg):
    cfg = pkg.synth.config("tiny", max_len=32)
    w = pkg.synth.make_model(cfg, O.Q4_K)
    toks = np.array([3, 77, 150, 9, 12], np.int32)
    a = O.Llama(cfg, w)
    la = a.forward(toks)
    b = O.Llama(cfg, w)                    # token by token must equal one prefill chunk (KV cache correctness)
    for t in toks:
        lb = b.forward([t])
    assert np.allclose(la, lb, rtol=0, atol=2e-4)
    assert np.isfinite(la).all() and la.std() > 0.05
