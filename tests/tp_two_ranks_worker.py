"""Worker of tests/test_gpu_tp.py::test_two_ranks_share_the_gpu_and_all_reduce_over_gloo: rank r of a 2-rank tensor-parallel group, BOTH ranks on GPU 0.
The runner's all-reduce goes through the host callback (cllm_llama_set_allreduce): device -> host, gloo all_reduce (CPU), host -> device -- so the HIP
tensor-parallel code of decoder.hip (sharded q/k/v/gate/up rows, o/down columns, partial sums folded into the next mat-vec's RMS_NORM prologue, residual
ping-pong) sees REAL partial sums from another rank.  argv: rank world port out.npz seed"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker_cfg(pkg):
    """TP_WORKER_CFG=uneven: Qwen2-style block (q/k/v biases, NEOX RoPE) whose ffn is NOT a multiple of 256 -- the reference then keeps down_proj in Q8_0
    (convert.py:811-829), 2848 / 32 = 89 blocks: rank 0 holds 45, rank 1 holds 44 (BASELINE cfg4's 924 blocks over 8 ranks in small)"""
    if os.environ.get("TP_WORKER_CFG") == "uneven":
        return pkg.synth.config("small", max_len=64, ffn=2848, qkv_bias=1, rope_mode=2)
    return pkg.synth.config("small", max_len=64, ffn=3072)          # 8 heads / 2 kv heads, ffn / 256 divisible by the group size


def main():
    rank, world, port, out, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_package
    import bench
    pkg = load_package()
    pkg.lib.require_gpu()
    L = pkg.lib.get()
    cfg = worker_cfg(pkg)
    w = pkg.synth.make_model(cfg, pkg.Q4_K, seed=seed)
    hd = cfg["head_dim"]
    QD, F = cfg["n_head"] * hd, cfg["ffn"]
    f0, fl = bench.ffn_share(cfg, pkg.synth.down_type(cfg, pkg.Q4_K), rank, world, pkg)      # whole quant blocks of the down projection: uneven where they do not divide
    sh = {}
    for name, (t, arr) in w.items():
        base = name.split(".")[-1]
        if base in ("wq", "wk", "wv", "bq", "bk", "bv"):
            arr = bench.shard_rows(arr, rank, world)
        elif base in ("wgate", "wup"):
            arr = np.ascontiguousarray(arr[f0:f0 + fl])
        elif base == "wo":
            arr = bench.shard_cols(arr, t, QD, rank, world, pkg)
        elif base == "wdown":
            arr = bench.shard_cols(arr, t, F, rank, world, pkg)
        sh[name] = (t, arr)
    m = pkg.Llama(cfg, sh, tp_rank=rank, tp_size=world, ffn_local=fl)
    n_calls = [0]

    def allreduce(stream, buf, n):
        pkg.ops.sync()                                         # the runner's launches are stream-ordered: finish them, then the host round trip
        host = np.empty(n, np.float32)
        pkg.lib.check(L.cllm_memcpy_d2h(host.ctypes.data_as(C.c_void_p), C.c_void_p(buf), n * 4, None), "d2h")
        t = torch.from_numpy(host)
        dist.all_reduce(t)
        pkg.lib.check(L.cllm_memcpy_h2d(C.c_void_p(buf), host.ctypes.data_as(C.c_void_p), n * 4, None), "h2d")
        pkg.ops.sync()
        n_calls[0] += 1
    oneshot = None
    if os.environ.get("TP_WORKER_MODE") == "oneshot":
        # the one-shot direct-write all-reduce (tp_oneshot.hip): both processes map each other's receive buffer through HIP IPC; the 64-byte handles travel over gloo
        oneshot = C.c_void_p()
        mine = (C.c_char * 64)()
        pkg.lib.check(L.cllm_tp_oneshot_create(rank, world, cfg["hidden"] * 64, C.byref(oneshot), mine), "oneshot_create")
        gathered = [None] * world
        dist.all_gather_object(gathered, bytes(mine.raw))
        pkg.lib.check(L.cllm_tp_oneshot_connect(oneshot, b"".join(gathered)), "oneshot_connect")
        dist.barrier()
        m.set_tp_oneshot(oneshot)
    else:
        m.set_allreduce(allreduce)
    fused = None
    if os.environ.get("TP_WORKER_MODE") == "fused":
        # the all-reduce fused into the mat-vecs (gemv_tp.hip): prompts and the node-by-node steps keep the gloo callback, the fused single-token steps have no
        # all-reduce launch at all -- the o / down launches write granules into both processes' receive buffers, the next RMS_NORM launch gathers them
        fused = C.c_void_p()
        mine = (C.c_char * 64)()
        pkg.lib.check(L.cllm_tp_fused_create(rank, world, 2 * cfg["n_layer"], cfg["hidden"], C.byref(fused), mine), "fused_create")
        gathered = [None] * world
        dist.all_gather_object(gathered, bytes(mine.raw))
        pkg.lib.check(L.cllm_tp_fused_connect(fused, b"".join(gathered)), "fused_connect")
        dist.barrier()
        m.set_tp_fused(fused)
    prompt = np.random.default_rng(seed).integers(0, cfg["vocab"], 12).astype(np.int32)
    teacher = np.random.default_rng(seed + 1).integers(0, cfg["vocab"], 10).astype(np.int32)
    logits = [m.forward(prompt)]
    for i, t in enumerate(teacher):                          # teacher-forced: the node-by-node path and the fused single-token path alternate
        logits.append(m.forward([int(t)]) if i % 2 == 0 else m.decode_fused_logits(int(t)))
    ids = m.decode_greedy(int(np.argmax(logits[-1])), 8)      # free-running through the fused TP step (eager launches: a host callback cannot be captured)
    err = L.cllm_tp_oneshot_error(oneshot) if oneshot else 0
    ferr = L.cllm_tp_fused_error(fused) if fused else 0
    if rank == 0:
        np.savez(out, logits=np.stack(logits), ids=ids, calls=n_calls[0], oneshot_error=err, fused_error=ferr)
    dist.barrier()                                           # (nobody unmaps a buffer a peer may still write)
    m.close()
    if oneshot:
        pkg.lib.check(L.cllm_tp_oneshot_destroy(oneshot), "oneshot_destroy")
    if fused:
        pkg.lib.check(L.cllm_tp_fused_destroy(fused), "fused_destroy")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
