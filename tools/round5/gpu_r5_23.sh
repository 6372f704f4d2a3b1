#!/bin/bash
# round 5, call 23: where the lm_head launch (k_gemv_rows, LDS-DMA ring) spends its 62 us: stream only / compute only / ring depth
O=gpurun_out/r5_23; mkdir -p $O
for v in base nocomp nodma ns2 ns4 base; do
  lib=$PWD/chatllm.cpp_amd/libchatllm_hip.so; [ $v != base ] && lib=$PWD/chatllm.cpp_amd/libchatllm_hip_$v.so
  CLLM_LIB=$lib timeout 300 python tools/gemv_bench.py --fused --types q4_k --iters 128 --shapes lm_head 2>&1 | grep fused | sed "s/^/$v /" | tee -a $O/summary.txt
done
python - <<'PY' 2>&1 | tee -a gpurun_out/r5_23/summary.txt
import sys; sys.path.insert(0, '.')
import __graft_entry__ as ge, ctypes as C
pkg = ge.load_package(); L = pkg.lib.get(); pkg.lib.require_gpu()
g = C.c_float()
pkg.lib.check(L.cllm_bench_read_bw(None, 2 << 30, 6, C.byref(g)), "read_bw"); print("streaming read ceiling %.2f TB/s" % (g.value / 1e3))
PY
