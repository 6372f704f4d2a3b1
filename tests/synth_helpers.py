"""random reference-format quant blocks for tests (numpy only; mirrors chatllm.cpp_amd/synth.py but fully random
scale bytes so that every bit pattern of the 6-bit scale packing is exercised)"""
import numpy as np

TYPE_SIZE = {2: 18, 3: 20, 8: 34, 12: 144, 13: 176, 14: 210}
BLCK = {2: 32, 3: 32, 8: 32, 12: 256, 13: 256, 14: 256}


def rand_blocks(t, rows, K, rng, d_scale=0.01):
    nb = K // BLCK[t]
    out = rng.integers(0, 256, (rows, nb, TYPE_SIZE[t]), dtype=np.uint8)
    d = (rng.uniform(0.25, 1.0, (rows, nb)) * d_scale).astype(np.float16)
    if t == 14:         # Q6_K: the fp16 scale closes the block (ql[128] qh[64] scales[16] d); keep the int8 sub-scales moderate
        out[:, :, 208:210] = d.view(np.uint8).reshape(rows, nb, 2)
        return np.ascontiguousarray(out.reshape(rows, nb * TYPE_SIZE[t]))
    out[:, :, 0:2] = d.view(np.uint8).reshape(rows, nb, 2)
    if t == 3:          # Q4_1: fp16 minimum after the scale (w = nib * d + m), both signs
        m = (rng.uniform(-8.0, 2.0, (rows, nb)) * d_scale).astype(np.float16)
        out[:, :, 2:4] = m.view(np.uint8).reshape(rows, nb, 2)
    if t in (12, 13):   # Q4_K / Q5_K: fp16 dmin
        dm = (rng.uniform(0.25, 1.0, (rows, nb)) * d_scale).astype(np.float16)
        out[:, :, 2:4] = dm.view(np.uint8).reshape(rows, nb, 2)
    return np.ascontiguousarray(out.reshape(rows, nb * TYPE_SIZE[t]))
