#!/usr/bin/env python3
"""Library GEMM ceilings on this box (context for the prefill MFMA fraction): torch.matmul (hipBLASLt / rocBLAS) at
8192^3 in fp16 / bf16 and torch._int_mm in int8."""
import time
import torch

dev = "cuda:0"
n = 8192
for name, mk in (("fp16", lambda: torch.randn(n, n, device=dev, dtype=torch.float16)), ("bf16", lambda: torch.randn(n, n, device=dev, dtype=torch.bfloat16))):
    a, b = mk(), mk()
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        c = a @ b
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"torch.matmul {name} {n}^3: {dt*1e3:.2f} ms  {2*n**3/dt/1e12:.0f} TFLOP/s")
try:
    a = torch.randint(-8, 8, (n, n), device=dev, dtype=torch.int8)
    b = torch.randint(-8, 8, (n, n), device=dev, dtype=torch.int8)
    for _ in range(3):
        c = torch._int_mm(a, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        c = torch._int_mm(a, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"torch._int_mm int8 {n}^3: {dt*1e3:.2f} ms  {2*n**3/dt/1e12:.0f} TOP/s")
except Exception as e:  # noqa: BLE001
    print("int8 library GEMM not available:", e)
