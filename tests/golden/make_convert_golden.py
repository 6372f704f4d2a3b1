#!/usr/bin/env python3
"""Golden vectors from the reference's OWN converter (convert.py:328-568: the torch quantizers that write every stock chatllm model file):
seeded float rows -> the bytes quantize_q8_0 / quantize_q4_0 / quantize_q4_1 / quantize_q4_k emit.  A second, independent statement of the weight encodings
(SURVEY 8c): tests/test_golden.py::test_convert_py_quantizers compares them with the oracle's from_float_ref restatements (which are pinned byte for byte to
libggml-base.so in tests/test_oracle_vs_reference.py).  Run in the BUILD container (needs /root/reference and torch); the .npz is committed."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import convert as C  # noqa: E402

rng = np.random.default_rng(20260924)
out = {}
for K in (256, 4096):
    for scale in (1.0, 1e-3, 40.0):
        x = (rng.standard_normal((6, K)) * scale).astype(np.float32)
        x[0, :32] = 0.0                      # an all-zero block
        x[1, 32:64] = 0.37                   # a constant block
        x[2, 64:96] = np.abs(x[2, 64:96]) + 0.1
        x[3, 100] = -x[3, 101]               # a tie in |x|
        key = f"K{K}_s{scale:g}"
        out["x_" + key] = x
        t = torch.from_numpy(x)
        out["q8_0_" + key] = C.quantize_q8_0(t).contiguous().view(torch.uint8).numpy().reshape(6, -1)
        out["q4_0_" + key] = C.quantize_q4_0(t).contiguous().view(torch.uint8).numpy().reshape(6, -1)
        out["q4_1_" + key] = C.quantize_q4_1(t).contiguous().view(torch.uint8).numpy().reshape(6, -1)
        out["q4_k_" + key] = C.quantize_q4_k(t, 256).contiguous().view(torch.uint8).numpy().reshape(6, -1)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "convert_py_reference.npz"), **out)
print({k: v.shape for k, v in out.items() if not k.startswith("x_")})
