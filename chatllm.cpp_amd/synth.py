"""Synthetic models at real shapes: weights are generated directly as reference-format quant
blocks (no float master, no network), SURVEY.md 8d.  Deterministic per tensor name, so any
subset (one layer for the CPU baseline, a tensor-parallel shard) can be regenerated identically.

Block layouts: /root/reference/ggml/src/ggml-common.h:170-175 (Q4_0), 177-189 (Q4_1), 219-224 (Q8_0), 295-306 (Q4_K).
"""
import zlib

import numpy as np

from .tensor import F32, Q4_0, Q4_1, Q8_0, Q4_K, Q5_K, Q6_K, Q5_0, Q5_1, Q2_K, Q3_K, IQ4_NL, MXFP4, IQ4_XS, TQ1_0, TQ2_0, IQ2_XXS, IQ2_XS, IQ2_S, IQ3_XXS, IQ3_S, IQ1_S, IQ1_M, TYPE_SIZE, BLCK  # noqa: F401

CONFIGS = {
    # name: n_layer, hidden, n_head, n_kv_head, head_dim, ffn, vocab
    "tiny":        dict(n_layer=2,  hidden=256,  n_head=4,  n_kv_head=2, head_dim=64,  ffn=512,   vocab=320),
    "small":       dict(n_layer=4,  hidden=1024, n_head=8,  n_kv_head=2, head_dim=128, ffn=2816,  vocab=2048),   # ffn % 256 == 0
    "gpt2s-llama": dict(n_layer=12, hidden=768,  n_head=12, n_kv_head=12, head_dim=64, ffn=3072,  vocab=50304),  # BASELINE cfg1 stand-in
    "llama3-8b":   dict(n_layer=32, hidden=4096, n_head=32, n_kv_head=8, head_dim=128, ffn=14336, vocab=128256),
    "mixtral-8x7b": dict(n_layer=32, hidden=4096, n_head=32, n_kv_head=8, head_dim=128, ffn=14336, vocab=32000, rope_theta=1e6),   # + 8 experts, top 2 (tools/make_ggmm.py --arch mixtral)
    "qwen2-72b":   dict(n_layer=80, hidden=8192, n_head=64, n_kv_head=8, head_dim=128, ffn=29568, vocab=152064, qkv_bias=1, rope_mode=2, rope_theta=1e6),
}


def config(name, max_len=1024, **over):
    c = dict(rope_mode=0, rope_theta=500000.0, rms_eps=1e-5, qkv_bias=0, max_len=max_len)
    c.update(CONFIGS[name])
    c.update(over)
    return c


def _rng(seed, name):
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def _f16_bytes(x):
    return np.asarray(x, np.float16).view(np.uint8)


def quant_blocks(type_, rows, K, rng, sigma):
    """random already-quantized rows [rows][K/blk] with dequantized std ~ sigma; returns uint8 [rows, row_bytes]"""
    nb = K // BLCK[type_]
    if type_ == Q8_0:
        out = np.empty((rows, nb, 34), np.uint8)
        d = rng.uniform(0.5, 1.5, (rows, nb)) * sigma / 42.0            # int8 ~ N(0, 42^2)
        out[:, :, 0:2] = _f16_bytes(d).reshape(rows, nb, 2)
        q = np.clip(np.rint(rng.standard_normal((rows, nb, 32), np.float32) * 42.0), -127, 127).astype(np.int8)
        out[:, :, 2:] = q.view(np.uint8)
    elif type_ == Q4_0:
        out = np.empty((rows, nb, 18), np.uint8)
        d = rng.uniform(0.5, 1.5, (rows, nb)) * sigma / 2.5             # (nib - 8) ~ N(0, 2.5^2)
        out[:, :, 0:2] = _f16_bytes(d).reshape(rows, nb, 2)
        nib = np.clip(np.rint(rng.standard_normal((rows, nb, 32), np.float32) * 2.5 + 8.0), 0, 15).astype(np.uint8)
        out[:, :, 2:] = nib[:, :, :16] | (nib[:, :, 16:] << 4)
    elif type_ == Q4_1:
        out = np.empty((rows, nb, 20), np.uint8)
        d = rng.uniform(0.5, 1.5, (rows, nb)) * sigma / 2.5             # w = nib * d + m with m = -7.5 d .. -8.5 d: ~zero-mean
        out[:, :, 0:2] = _f16_bytes(d).reshape(rows, nb, 2)
        out[:, :, 2:4] = _f16_bytes(-d * rng.uniform(7.0, 9.0, (rows, nb))).reshape(rows, nb, 2)
        nib = np.clip(np.rint(rng.standard_normal((rows, nb, 32), np.float32) * 2.5 + 8.0), 0, 15).astype(np.uint8)
        out[:, :, 4:] = nib[:, :, :16] | (nib[:, :, 16:] << 4)
    elif type_ == Q4_K:
        out = np.empty((rows, nb, 144), np.uint8)
        sc = rng.integers(20, 64, (rows, nb, 8), dtype=np.uint8)        # 6-bit sub-block scales
        # w = d*sc*q - dmin*m ; choose dmin = 8 d and m ~ sc*7.5/8 so that the weights are ~zero-mean
        m = np.clip(np.rint(sc.astype(np.float32) * (7.5 / 8.0)), 0, 63).astype(np.uint8)
        d = rng.uniform(0.5, 1.5, (rows, nb)) * sigma / (42.0 * 2.5)
        out[:, :, 0:2] = _f16_bytes(d).reshape(rows, nb, 2)
        out[:, :, 2:4] = _f16_bytes(d * 8.0).reshape(rows, nb, 2)
        s = out[:, :, 4:16]                                             # inverse of get_scale_min_k4 (ggml-quants.c:703-711)
        s[:, :, 0:4] = (sc[:, :, 0:4] & 63) | ((sc[:, :, 4:8] >> 4) << 6)
        s[:, :, 4:8] = (m[:, :, 0:4] & 63) | ((m[:, :, 4:8] >> 4) << 6)
        s[:, :, 8:12] = (sc[:, :, 4:8] & 0xF) | ((m[:, :, 4:8] & 0xF) << 4)
        nib = np.clip(np.rint(rng.standard_normal((rows, nb, 256), np.float32) * 2.5 + 7.5), 0, 15).astype(np.uint8)
        nib = nib.reshape(rows, nb, 4, 2, 32)                           # 4 groups of 64: low nibbles then high nibbles
        out[:, :, 16:] = (nib[:, :, :, 0, :] | (nib[:, :, :, 1, :] << 4)).reshape(rows, nb, 128)
    elif type_ == Q5_K:                                                 # Q4_K's header + a fifth bit per weight (ggml-common.h:308-321)
        out = np.empty((rows, nb, 176), np.uint8)
        sc = rng.integers(20, 64, (rows, nb, 8), dtype=np.uint8)
        m = np.clip(np.rint(sc.astype(np.float32) * (15.5 / 8.0)), 0, 63).astype(np.uint8)          # w = d*sc*q - dmin*m, q ~ N(15.5, 5^2): ~zero-mean
        d = rng.uniform(0.5, 1.5, (rows, nb)) * sigma / (42.0 * 5.0)
        out[:, :, 0:2] = _f16_bytes(d).reshape(rows, nb, 2)
        out[:, :, 2:4] = _f16_bytes(d * 8.0).reshape(rows, nb, 2)
        s = out[:, :, 4:16]
        s[:, :, 0:4] = (sc[:, :, 0:4] & 63) | ((sc[:, :, 4:8] >> 4) << 6)
        s[:, :, 4:8] = (m[:, :, 0:4] & 63) | ((m[:, :, 4:8] >> 4) << 6)
        s[:, :, 8:12] = (sc[:, :, 4:8] & 0xF) | ((m[:, :, 4:8] & 0xF) << 4)
        q = np.clip(np.rint(rng.standard_normal((rows, nb, 256), np.float32) * 5.0 + 15.5), 0, 31).astype(np.uint8)
        q = q.reshape(rows, nb, 4, 2, 32)                               # 4 groups of 64: low nibbles then high nibbles; bit 4 -> qh bit 2g + half
        out[:, :, 48:] = ((q[:, :, :, 0, :] & 15) | ((q[:, :, :, 1, :] & 15) << 4)).reshape(rows, nb, 128)
        qh = np.zeros((rows, nb, 32), np.uint8)
        for g in range(4):
            qh |= ((q[:, :, g, 0, :] >> 4) << (2 * g)) | ((q[:, :, g, 1, :] >> 4) << (2 * g + 1))
        out[:, :, 16:48] = qh
    elif type_ == Q6_K:                                                 # ql[128] qh[64] scales[16] d (ggml-common.h:323-336): w = d * sc * (q - 32)
        out = np.empty((rows, nb, 210), np.uint8)
        sc = rng.integers(40, 128, (rows, nb, 16)).astype(np.int8) * rng.choice(np.array([1, 1, 1, -1], np.int8), (rows, nb, 16))
        d = rng.uniform(0.5, 1.5, (rows, nb)) * sigma / (84.0 * 10.0)
        out[:, :, 208:210] = _f16_bytes(d).reshape(rows, nb, 2)
        out[:, :, 192:208] = sc.view(np.uint8)
        q = np.clip(np.rint(rng.standard_normal((rows, nb, 256), np.float32) * 10.0 + 32.0), 0, 63).astype(np.uint8)
        q = q.reshape(rows, nb, 2, 4, 32)                               # halves of 128: four groups of 32 (dequantize_row_q6_K, ggml-quants.c:1762-1791)
        ql = np.empty((rows, nb, 2, 64), np.uint8)
        ql[:, :, :, 0:32] = (q[:, :, :, 0, :] & 15) | ((q[:, :, :, 2, :] & 15) << 4)
        ql[:, :, :, 32:64] = (q[:, :, :, 1, :] & 15) | ((q[:, :, :, 3, :] & 15) << 4)
        out[:, :, 0:128] = ql.reshape(rows, nb, 128)
        qh = (q[:, :, :, 0, :] >> 4) | ((q[:, :, :, 1, :] >> 4) << 2) | ((q[:, :, :, 2, :] >> 4) << 4) | ((q[:, :, :, 3, :] >> 4) << 6)
        out[:, :, 128:192] = qh.reshape(rows, nb, 64)
    elif type_ in (Q5_0, Q5_1):                                         # d [m] qh[4] qs[16] (ggml-common.h:197-216): w = (q5 - 16) d | q5 d + m
        off = 2 if type_ == Q5_0 else 4
        out = np.empty((rows, nb, TYPE_SIZE[type_]), np.uint8)
        d = rng.uniform(0.5, 1.5, (rows, nb)) * sigma / 5.0
        out[:, :, 0:2] = _f16_bytes(d).reshape(rows, nb, 2)
        if type_ == Q5_1:
            out[:, :, 2:4] = _f16_bytes(-d * rng.uniform(15.0, 17.0, (rows, nb))).reshape(rows, nb, 2)
        q = np.clip(np.rint(rng.standard_normal((rows, nb, 32), np.float32) * 5.0 + 16.0), 0, 31).astype(np.uint32)
        out[:, :, off + 4:] = ((q[:, :, :16] & 15) | ((q[:, :, 16:] & 15) << 4)).astype(np.uint8)
        qh = np.zeros((rows, nb), np.uint32)
        for e in range(32):
            qh |= (q[:, :, e] >> 4) << e
        out[:, :, off:off + 4] = qh.view(np.uint8).reshape(rows, nb, 4)
    elif type_ in (IQ4_NL, MXFP4):                                      # d | e, qs[16]: 16-entry int8 codebooks (ggml-common.h:190-194, 415-419, 1088-1096)
        out = np.empty((rows, nb, TYPE_SIZE[type_]), np.uint8)
        off = 2 if type_ == IQ4_NL else 1
        if type_ == IQ4_NL:
            d = rng.uniform(0.5, 1.5, (rows, nb)) * sigma / 72.0        # kvalues_iq4nl: rms 72
            out[:, :, 0:2] = _f16_bytes(d).reshape(rows, nb, 2)
        else:
            e = np.clip(np.rint(128.0 + np.log2(max(sigma, 1e-30) / 5.7) + rng.uniform(-0.5, 0.5, (rows, nb))), 2, 250)      # kvalues_mxfp4: rms 5.7, scale 2^(e - 128)
            out[:, :, 0] = e.astype(np.uint8)
        out[:, :, off:] = rng.integers(0, 256, (rows, nb, 16), dtype=np.uint8)
    elif type_ == IQ4_XS:                                               # d scales_h scales_l[4] qs[128] (ggml-common.h:421-427): w = d (ls - 32) kvalues_iq4nl[nib], ls = 6 bits per 32
        out = np.empty((rows, nb, 136), np.uint8)
        d = rng.uniform(0.5, 1.5, (rows, nb)) * sigma / (72.0 * 18.0)   # codebook rms 72, |ls - 32| rms ~18
        out[:, :, 0:2] = _f16_bytes(d).reshape(rows, nb, 2)
        out[:, :, 2:8] = rng.integers(0, 256, (rows, nb, 6), dtype=np.uint8)          # every 6-bit scale pattern, both signs of ls - 32
        out[:, :, 8:] = rng.integers(0, 256, (rows, nb, 128), dtype=np.uint8)
    elif type_ in (IQ2_XXS, IQ2_XS, IQ2_S, IQ3_XXS, IQ3_S):             # the codebook formats (ggml-common.h:346-390): fp16 d, then indices / signs / scales -- every bit pattern is a valid block
        out = rng.integers(0, 256, (rows, nb, TYPE_SIZE[type_]), dtype=np.uint8)
        rms = {IQ2_XXS: 29.0 * 2.0, IQ2_XS: 29.0 * 2.0, IQ2_S: 29.0 * 2.0, IQ3_XXS: 33.0 * 4.0, IQ3_S: 9.2 * 16.0}[type_]        # magnitude rms x mean scale factor
        out[:, :, 0:2] = _f16_bytes(rng.uniform(0.5, 1.5, (rows, nb)) * sigma / rms).reshape(rows, nb, 2)
    elif type_ in (IQ1_S, IQ1_M):                                       # values -1 / 0 / 1 (+- 1/8) times d (2 s + 1), s in 0..7: rms ~0.83 * 9; IQ1_M keeps its fp16 d in the scales' top nibbles
        out = rng.integers(0, 256, (rows, nb, TYPE_SIZE[type_]), dtype=np.uint8)
        dv = _f16_bytes(rng.uniform(0.5, 1.5, (rows, nb)) * sigma / 7.5).reshape(rows, nb, 2)
        if type_ == IQ1_S:
            out[:, :, 0:2] = dv
        else:
            d16 = dv.copy().view(np.uint16).reshape(rows, nb).astype(np.uint32)
            sc = out[:, :, 48:56].copy().view(np.uint16).reshape(rows, nb, 4)
            for k in range(4):
                sc[:, :, k] = (sc[:, :, k] & 0x0fff) | (((d16 >> (4 * k)) & 0xf) << 12).astype(np.uint16)
            out[:, :, 48:56] = sc.view(np.uint8).reshape(rows, nb, 8)
    elif type_ == TQ2_0:                                                # qs[64] d (ggml-common.h:251-256): w = (q - 1) d, q in 0..2 at 2 bits
        out = np.empty((rows, nb, 66), np.uint8)
        q = rng.integers(0, 3, (rows, nb, 64, 4), dtype=np.uint8)
        out[:, :, :64] = q[..., 0] | (q[..., 1] << 2) | (q[..., 2] << 4) | (q[..., 3] << 6)
        out[:, :, 64:66] = _f16_bytes(rng.uniform(0.5, 1.5, (rows, nb)) * sigma / 0.816).reshape(rows, nb, 2)
    elif type_ == TQ1_0:                                                # qs[48] qh[4] d (ggml-common.h:241-249): five trits per byte as ceil(base-3 value * 256 / 243), four in qh
        out = np.empty((rows, nb, 54), np.uint8)
        out[:, :, :48] = ((rng.integers(0, 243, (rows, nb, 48)) * 256 + 242) // 243).astype(np.uint8)
        out[:, :, 48:52] = ((rng.integers(0, 81, (rows, nb, 4)) * 3 * 256 + 242) // 243).astype(np.uint8)
        out[:, :, 52:54] = _f16_bytes(rng.uniform(0.5, 1.5, (rows, nb)) * sigma / 0.816).reshape(rows, nb, 2)
    elif type_ == Q2_K:                                                 # scales[16] (scale | min << 4) qs[64] d dmin: w = d sc q - dmin m, q in 0..3
        out = np.empty((rows, nb, 84), np.uint8)
        sc = rng.integers(3, 11, (rows, nb, 16), dtype=np.uint8)
        m = np.clip(np.rint(sc.astype(np.float32) * 1.5), 0, 15).astype(np.uint8)       # dmin = d: ~zero-mean
        out[:, :, 0:16] = sc | (m << 4)
        out[:, :, 16:80] = rng.integers(0, 256, (rows, nb, 64), dtype=np.uint8)
        d = rng.uniform(0.5, 1.5, (rows, nb)) * sigma / (7.0 * 1.12)
        out[:, :, 80:82] = _f16_bytes(d).reshape(rows, nb, 2)
        out[:, :, 82:84] = _f16_bytes(d).reshape(rows, nb, 2)
    elif type_ == Q3_K:                                                 # hmask[32] qs[64] scales[12] d: w = d (sc - 32) (q2 | h << 2) - 4): q in -4..3
        out = np.empty((rows, nb, 110), np.uint8)
        out[:, :, 0:96] = rng.integers(0, 256, (rows, nb, 96), dtype=np.uint8)
        sc = (32 + rng.integers(8, 32, (rows, nb, 16)) * rng.choice(np.array([1, 1, 1, -1]), (rows, nb, 16))).astype(np.uint8)      # 6-bit codes, value - 32 in +-(8..31)
        s = np.zeros((rows, nb, 12), np.uint8)                          # inverse of the unpack of ggml-quants.c:1141-1150
        s[:, :, 0:8] = (sc[:, :, 0:8] & 15) | ((sc[:, :, 8:16] & 15) << 4)
        for k in range(4):
            s[:, :, 8:12] |= ((sc[:, :, 4 * k:4 * k + 4] >> 4) & 3) << (2 * k)
        out[:, :, 96:108] = s
        d = rng.uniform(0.5, 1.5, (rows, nb)) * sigma / (20.0 * 2.3)
        out[:, :, 108:110] = _f16_bytes(d).reshape(rows, nb, 2)
    else:
        raise ValueError(type_)
    return out.reshape(rows, nb * TYPE_SIZE[type_])


def down_type(cfg, wtype):
    """the reference falls back to Q8_0 when a row is not a multiple of 256 (convert.py:811-829, src/layers.cpp:81-95)"""
    return Q8_0 if (wtype in (Q4_K, Q5_K, Q6_K) and cfg["ffn"] % 256) else wtype


def tensor_list(cfg, wtype, tp_rank=0, tp_size=1):
    """[(name, type, rows, K, row_slice)] of one (shard of a) model; row_slice selects the tensor-parallel rows"""
    H, hd, F, V = cfg["hidden"], cfg["head_dim"], cfg["ffn"], cfg["vocab"]
    QD, KD = cfg["n_head"] * hd, cfg["n_kv_head"] * hd
    mix = cfg.get("mix") or {}                  # per-tensor types by suffix, like the Q4_K_M / Q5_K_M mixes of third-party files: {"wv": Q6_K, "lm_head": Q6_K}
    t = lambda n: mix.get(n, wtype)             # noqa: E731
    out = [("tok_embd", t("tok_embd"), V, H), ("lm_head", t("lm_head"), V, H)]
    for i in range(cfg["n_layer"]):
        p = f"layers.{i}."
        out += [(p + "wq", t("wq"), QD, H), (p + "wk", t("wk"), KD, H), (p + "wv", t("wv"), KD, H), (p + "wo", t("wo"), H, QD),
                (p + "wgate", t("wgate"), F, H), (p + "wup", t("wup"), F, H), (p + "wdown", down_type(cfg, t("wdown")), H, F)]
    return out


def make_tensor(name, type_, rows, K, seed=1234):
    return quant_blocks(type_, rows, K, _rng(seed, name), 1.0 / np.sqrt(K))


def make_norm(name, H, seed=1234):
    return (1.0 + 0.02 * _rng(seed, name).standard_normal(H)).astype(np.float32)


def make_bias(name, n, seed=1234):
    return (0.02 * _rng(seed, name).standard_normal(n)).astype(np.float32)


def make_model(cfg, wtype, seed=1234, layers=None):
    """dict name -> (type, numpy array) for the whole model (or only `layers`); small configs only on the host"""
    w = {}
    keep = None if layers is None else {f"layers.{i}." for i in layers}
    for name, t, rows, K in tensor_list(cfg, wtype):
        if keep is not None and name.startswith("layers.") and not any(name.startswith(k) for k in keep):
            continue
        w[name] = (t, make_tensor(name, t, rows, K, seed))
    H = cfg["hidden"]
    w["out_norm"] = (F32, make_norm("out_norm", H, seed))
    for i in (range(cfg["n_layer"]) if layers is None else layers):
        p = f"layers.{i}."
        w[p + "attn_norm"] = (F32, make_norm(p + "attn_norm", H, seed))
        w[p + "ffn_norm"] = (F32, make_norm(p + "ffn_norm", H, seed))
        if cfg.get("qkv_bias"):
            hd = cfg["head_dim"]
            w[p + "bq"] = (F32, make_bias(p + "bq", cfg["n_head"] * hd, seed))
            w[p + "bk"] = (F32, make_bias(p + "bk", cfg["n_kv_head"] * hd, seed))
            w[p + "bv"] = (F32, make_bias(p + "bv", cfg["n_kv_head"] * hd, seed))
    return w


def weight_bytes_per_token(cfg, wtype):
    """algorithmic weight bytes one decoded token touches: all layer matrices + lm_head + norms"""
    total = 0
    for name, t, rows, K in tensor_list(cfg, wtype):
        if name == "tok_embd":
            continue
        total += rows * (K // BLCK[t]) * TYPE_SIZE[t]
    total += (2 * cfg["n_layer"] + 1) * cfg["hidden"] * 4
    return total


def kv_bytes_per_token(cfg, n_ctx):
    """F16 K and V read for n_ctx cached positions + one position written, all layers"""
    kd = cfg["n_kv_head"] * cfg["head_dim"]
    return cfg["n_layer"] * 2 * kd * 2 * (n_ctx + 1)


def fast_blocks(type_, rows, K, rng, sigma):
    """bench-size generator (GB/s instead of MB/s): uniformly random payload bytes, sane fp16 scale fields.
    Dequantized std ~ sigma (uniform nibbles / int8 instead of rounded normals)."""
    nb = K // BLCK[type_]
    bs = TYPE_SIZE[type_]
    out = rng.integers(0, 256, (rows, nb, bs), dtype=np.uint8)
    # one random fp16 scale per block from a small table (cheap): +-25 % around the target
    if type_ == Q8_0:
        base = sigma / 73.9            # uniform int8 std
    elif type_ in (Q4_0, Q4_1):
        base = sigma / 4.61            # uniform nibble std
    else:
        base = sigma / (4.61 * 32.0)   # nibble std x mean 6-bit scale
    if type_ == Q6_K:
        base = sigma / (18.5 * 74.0)   # 6-bit std x int8 scale std; the fp16 scale closes the block
    table = (base * np.linspace(0.75, 1.25, 256)).astype(np.float16).view(np.uint16)
    sel = out[:, :, 0].copy()
    d = table[sel]
    if type_ == Q6_K:
        out[:, :, 208] = (d & 0xff).astype(np.uint8)
        out[:, :, 209] = (d >> 8).astype(np.uint8)
        return out.reshape(rows, nb * bs)
    out[:, :, 0] = (d & 0xff).astype(np.uint8)
    out[:, :, 1] = (d >> 8).astype(np.uint8)
    if type_ == Q4_1:
        dm = (-(base * 7.5) * np.linspace(0.75, 1.25, 256)).astype(np.float16).view(np.uint16)[sel]      # m = -7.5 d: zero-mean weights
        out[:, :, 2] = (dm & 0xff).astype(np.uint8)
        out[:, :, 3] = (dm >> 8).astype(np.uint8)
    if type_ in (Q4_K, Q5_K):
        # dmin = 7.5 d: with independent 6-bit scales and mins the sub-block offsets average out
        dm = (base * 7.5 * np.linspace(0.75, 1.25, 256)).astype(np.float16).view(np.uint16)[sel]
        out[:, :, 2] = (dm & 0xff).astype(np.uint8)
        out[:, :, 3] = (dm >> 8).astype(np.uint8)
    return out.reshape(rows, nb * bs)


def make_tensor_fast(name, type_, rows, K, seed=1234):
    return fast_blocks(type_, rows, K, _rng(seed, name), 1.0 / np.sqrt(K))
