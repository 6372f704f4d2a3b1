"""random reference-format quant blocks for tests (numpy only; mirrors chatllm.cpp_amd/synth.py but fully random
scale bytes so that every bit pattern of the 6-bit scale packing is exercised)"""
import numpy as np

TYPE_SIZE = {2: 18, 3: 20, 8: 34, 12: 144, 13: 176, 14: 210, 6: 22, 7: 24, 10: 84, 11: 110, 20: 18, 39: 17, 23: 136, 34: 54, 35: 66, 16: 66, 17: 74, 22: 82, 18: 98, 21: 110, 19: 50, 29: 56}
BLCK = {2: 32, 3: 32, 8: 32, 12: 256, 13: 256, 14: 256, 6: 32, 7: 32, 10: 256, 11: 256, 20: 32, 39: 32, 23: 256, 34: 256, 35: 256, 16: 256, 17: 256, 22: 256, 18: 256, 21: 256, 19: 256, 29: 256}


def rand_blocks(t, rows, K, rng, d_scale=0.01):
    nb = K // BLCK[t]
    out = rng.integers(0, 256, (rows, nb, TYPE_SIZE[t]), dtype=np.uint8)
    d = (rng.uniform(0.25, 1.0, (rows, nb)) * d_scale).astype(np.float16)
    if t == 14:         # Q6_K: the fp16 scale closes the block (ql[128] qh[64] scales[16] d); keep the int8 sub-scales moderate
        out[:, :, 208:210] = d.view(np.uint8).reshape(rows, nb, 2)
        return np.ascontiguousarray(out.reshape(rows, nb * TYPE_SIZE[t]))
    if t == 39:         # MXFP4: one E8M0 exponent byte (2^(e - 128)), then the 16 code bytes: exponents around the other types' scales, and the two denormal patterns
        out[:, :, 0] = rng.integers(114, 124, (rows, nb), dtype=np.uint8)
        out[0, : min(nb, 2), 0] = np.arange(min(nb, 2), dtype=np.uint8)
        return np.ascontiguousarray(out.reshape(rows, nb * TYPE_SIZE[t]))
    if t == 10:         # Q2_K: scales[16] qs[64] d dmin
        dm = (rng.uniform(0.25, 1.0, (rows, nb)) * d_scale).astype(np.float16)
        out[:, :, 80:82] = d.view(np.uint8).reshape(rows, nb, 2)
        out[:, :, 82:84] = dm.view(np.uint8).reshape(rows, nb, 2)
        return np.ascontiguousarray(out.reshape(rows, nb * TYPE_SIZE[t]))
    if t == 29:         # IQ1_M: no d field -- the fp16 scale lives in the top nibbles of the four 16-bit scale words (qs[32] qh[16] scales[8])
        sc = out[:, :, 48:56].copy().view(np.uint16).reshape(rows, nb, 4)
        dv = d.view(np.uint16).astype(np.uint32)
        for k in range(4):
            sc[:, :, k] = (sc[:, :, k] & 0x0fff) | (((dv >> (4 * k)) & 0xf) << 12).astype(np.uint16)
        out[:, :, 48:56] = sc.view(np.uint8).reshape(rows, nb, 8)
        return np.ascontiguousarray(out.reshape(rows, nb * TYPE_SIZE[t]))
    if t in (34, 35):   # TQ1_0: qs[48] qh[4] d (any byte decodes to trits 0..2); TQ2_0: qs[64] d, 2-bit values 0..2 (and a few 3s: "should not be", but the arithmetic is defined)
        off = 52 if t == 34 else 64
        if t == 35:
            q = rng.integers(0, 3, (rows, nb, 64, 4), dtype=np.uint8)
            q[rng.random((rows, nb, 64, 4)) < 0.01] = 3
            out[:, :, :64] = q[..., 0] | (q[..., 1] << 2) | (q[..., 2] << 4) | (q[..., 3] << 6)
        out[:, :, off:off + 2] = d.view(np.uint8).reshape(rows, nb, 2)
        return np.ascontiguousarray(out.reshape(rows, nb * TYPE_SIZE[t]))
    if t == 11:         # Q3_K: hmask[32] qs[64] scales[12] d
        out[:, :, 108:110] = d.view(np.uint8).reshape(rows, nb, 2)
        return np.ascontiguousarray(out.reshape(rows, nb * TYPE_SIZE[t]))
    out[:, :, 0:2] = d.view(np.uint8).reshape(rows, nb, 2)
    if t == 7:          # Q5_1: fp16 minimum after the scale
        m = (rng.uniform(-16.0, 4.0, (rows, nb)) * d_scale).astype(np.float16)
        out[:, :, 2:4] = m.view(np.uint8).reshape(rows, nb, 2)
    if t == 3:          # Q4_1: fp16 minimum after the scale (w = nib * d + m), both signs
        m = (rng.uniform(-8.0, 2.0, (rows, nb)) * d_scale).astype(np.float16)
        out[:, :, 2:4] = m.view(np.uint8).reshape(rows, nb, 2)
    if t in (12, 13):   # Q4_K / Q5_K: fp16 dmin
        dm = (rng.uniform(0.25, 1.0, (rows, nb)) * d_scale).astype(np.float16)
        out[:, :, 2:4] = dm.view(np.uint8).reshape(rows, nb, 2)
    return np.ascontiguousarray(out.reshape(rows, nb * TYPE_SIZE[t]))


def rms_boundary_rows(n, rows, rng):
    """rows whose mean of squares sits on a float ROUNDING BOUNDARY to ~2^-54, with terms spread over 36 binades so that the double accumulation rounds at
    every step: the float the reference's serial loop (ggml_compute_forward_rms_norm_f32, ops.cpp:3736-3741) rounds the mean to is then decided by the
    ORDER of the additions -- a tree and the serial loop disagree on about half of these rows.  Exact rational arithmetic picks the last two elements."""
    from fractions import Fraction
    out = np.zeros((rows, n), np.float32)
    for r in range(rows):
        x = (rng.standard_normal(n) * 2.0 ** -12).astype(np.float32)
        x[0] = np.float32(64.0)                                    # s0 = 4096: every later term loses its low bits to the accumulator's ulp (2^-40)
        x[n - 2] = x[n - 1] = 0.0
        sq = lambda v: Fraction(float(np.float32(v) * np.float32(v)))      # the float product the loop adds
        rest = sum(sq(v) for v in x[: n - 2])
        m = np.float32(float(rest / n) * (1 + 2.0 ** -10))         # a float a little above the mean so far ...
        target = (Fraction(float(m)) + Fraction(float(np.spacing(m))) / 2) * n      # ... and the midpoint to its successor, as a sum
        # coarse: the largest float whose float square stays below what is missing; fine: the same for the remainder (squares ~2^-30, spacing ~2^-54)
        for j in (n - 2, n - 1):
            miss = target - rest
            v = np.float32(np.sqrt(float(miss)))
            while sq(v) > miss:
                v = np.nextafter(v, np.float32(0))
            x[j] = v
            rest += sq(v)
        out[r] = x
    return out
