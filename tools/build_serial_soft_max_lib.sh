#!/bin/bash
# Test build: chatllm.cpp_amd/libchatllm_hip_serial.so = the library with SOFT_FORCE_SERIAL=1 -- every soft_max kernel ALWAYS redoes its double total in the reference's serial
# order (the path the per-row order proof of common.h falls back to about once in 2^20 rows).  Run the attention / soft_max / llama tests on it with
#   CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip_serial.so python -m pytest tests/test_gpu_ops.py tests/test_gpu_llama.py -m gpu -q
# (tools/round5/gpu_r5_35.sh: 96 + 45 passed on both libraries).
set -e
cd "$(dirname "$0")/../chatllm.cpp_amd/csrc"
make -j16 > /dev/null
mkdir -p build_serial
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fvisibility=hidden -mllvm -amdgpu-kernarg-preload-count=16"
for f in ops decode_fused attn_long gemv_moe; do /opt/rocm/bin/hipcc $F -DSOFT_FORCE_SERIAL=1 -c $f.hip -o build_serial/$f.o & done; wait
objs=""; for src in $(sed -n 's/^SRC *:= *//p' Makefile); do b=${src%.hip}.o; if [ -f build_serial/$b ]; then objs="$objs build_serial/$b"; else objs="$objs build/$b"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libchatllm_hip_serial.so $objs -ldl
echo "built ../libchatllm_hip_serial.so"
