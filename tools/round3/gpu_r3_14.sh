#!/bin/bash
O=gpurun_out/r3n; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "weight_quantizers" 2>&1 | tail -8 | tee $O/pytest_quant.txt
python - <<'PY' 2>&1 | tee $O/quantize_speed.txt
import ctypes as C, time, numpy as np, sys
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib.get(); T = pkg.Tensor
K, rows = 4096, 28672          # one gate/up matrix of Llama-3-8B: 117 M weights
x = T.from_numpy(np.random.default_rng(0).standard_normal((rows, K)).astype(np.float32))
for name, t, bpb, blk in (("q8_0", 8, 34, 32), ("q4_0", 2, 18, 32), ("q4_1", 3, 20, 32), ("q5_0", 6, 22, 32), ("q5_1", 7, 24, 32), ("q4_k", 12, 144, 256), ("f16", 1, 2, 1)):
    out = T(t, [K, rows])
    pkg.lib.check(L.cllm_op_quantize_rows(None, t, x.data_ptr(), out.data_ptr(), K, rows), "q"); pkg.lib.check(L.cllm_stream_sync(None), "s")
    t0 = time.perf_counter()
    for _ in range(3): pkg.lib.check(L.cllm_op_quantize_rows(None, t, x.data_ptr(), out.data_ptr(), K, rows), "q")
    pkg.lib.check(L.cllm_stream_sync(None), "s")
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {rows}x{K} fp32 -> {name} in {dt*1e3:.2f} ms = {rows*K/dt/1e9:.1f} G weights/s")
PY
cd oracle/_ref; M=/tmp/llama3-8b-q4k.bin
[ -s $M ] || python /root/repo/tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 512 --fast --out $M
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
for fa in 0 1; do
  for cache in f16 q8_0; do
    [ $fa = 0 ] && [ $cache = q8_0 ] && continue
    if [ $fa = 1 ]; then export REF_CHAT_FA=1; else unset REF_CHAT_FA; fi
    REF_CHAT_CACHE=$cache CLLM_HIP_STATS=1 ./ref_chat $M all 16 80 - $IDS > /tmp/fa_ids.txt 2> /tmp/fa_err.txt
    echo "--- fa=$fa cache=$cache"; grep "^decode:" /tmp/fa_err.txt; grep "per graph" /tmp/fa_err.txt | tail -1; grep "calls (" /tmp/fa_err.txt | tail -1
  done
done 2>&1 | cut -c1-600 | tee /root/repo/gpurun_out/r3n/fa_host_breakdown.txt
