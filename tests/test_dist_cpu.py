"""CPU-only coverage of the N>1 path: world_size-2 `gloo` processes shard one decoder layer the way bench.py shards it for
tensor parallelism (q/k/v/gate/up by output rows = heads / ffn columns, o/down by input columns = whole quant blocks), run the
oracle on their shard, all-reduce(sum) the partial o_proj / down_proj outputs, and must reproduce the unsharded layer.
(Activation quantization is block-local and shard boundaries are block-aligned, so the integer sums are identical; only the
fp32 order of the final reduction differs.)"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_package

torch = pytest.importorskip("torch")
import torch.distributed as dist          # noqa: E402
import torch.multiprocessing as mp        # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _layer(O, cfg, w, x, nh, nkv, F, shard=None):
    """attention-less slice of one layer that exercises every sharded mat-mul: returns (o_partial, down_partial)"""
    H, hd = cfg["hidden"], cfg["head_dim"]
    QD, KD = nh * hd, nkv * hd

    def mm(wt, K, N, inp):
        t, arr = wt
        out = np.zeros((1, N), np.float32)
        O.mul_mat(O.tensor(arr, t, [K, N]), O.tensor(np.ascontiguousarray(inp), O.F32, [K, 1]), O.tensor(out, O.F32, [N, 1]))
        return out
    q = mm(w["wq"], H, QD, x)
    v = mm(w["wv"], H, KD, x)
    att = q * np.repeat(v.reshape(nkv, hd), nh // nkv, axis=0).reshape(1, QD)      # a stand-in for attention that keeps the head structure
    o = mm(w["wo"], QD, H, att)
    g = mm(w["wgate"], H, F, x)
    u = mm(w["wup"], H, F, x)
    hcur = np.zeros_like(g)
    O.silu(O.tensor(g, O.F32, [F, 1]), O.tensor(hcur, O.F32, [F, 1]))
    d = mm(w["wdown"], F, H, hcur * u)
    return o, d


def _worker(rank, world, port, q_out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    import bench
    pkg = load_package()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # (Q8_0, 2848): 89 quant blocks in a down_proj row -- NOT divisible by the group: rank 0 holds 45 blocks, rank 1 holds 44 (bench.tp_split == cllm_tp_split),
        # the gate / up rows follow the blocks (BASELINE cfg4: Qwen2-72B's Q8_0 down_proj, 924 blocks over 8 ranks)
        for wtype, ffn in ((O.Q4_K, 3072), (O.Q8_0, 2816), (O.Q8_0, 2848)):
            cfg = dict(pkg.synth.config("small"), ffn=ffn)  # hd 128, 8 heads / 2 kv heads, hidden 1024
            H, hd, F = cfg["hidden"], cfg["head_dim"], cfg["ffn"]
            full = {k.split(".")[-1]: v for k, v in pkg.synth.make_model(cfg, wtype, seed=3, layers=[0]).items() if k.startswith("layers.0.")}
            x = np.random.default_rng(5).standard_normal((1, H)).astype(np.float32)
            QD = cfg["n_head"] * hd
            f0, fl = bench.ffn_share(cfg, full["wdown"][0], rank, world, pkg)
            sh = dict(full)
            for k in ("wq", "wk", "wv"):
                sh[k] = (full[k][0], bench.shard_rows(full[k][1], rank, world))
            for k in ("wgate", "wup"):
                sh[k] = (full[k][0], np.ascontiguousarray(full[k][1][f0:f0 + fl]))
            sh["wo"] = (full["wo"][0], bench.shard_cols(full["wo"][1], full["wo"][0], QD, rank, world, pkg))
            sh["wdown"] = (full["wdown"][0], bench.shard_cols(full["wdown"][1], full["wdown"][0], F, rank, world, pkg))
            assert sh["wdown"][1].shape[1] == fl // pkg.tensor.BLCK[full["wdown"][0]] * pkg.tensor.TYPE_SIZE[full["wdown"][0]]
            o_p, d_p = _layer(O, cfg, sh, x, cfg["n_head"] // world, cfg["n_kv_head"] // world, fl)
            to, td = torch.from_numpy(o_p.copy()), torch.from_numpy(d_p.copy())
            dist.all_reduce(to)                              # the residual-stream all-reduce of the north star
            dist.all_reduce(td)
            o_f, d_f = _layer(O, cfg, full, x, cfg["n_head"], cfg["n_kv_head"], F)
            eo = float(np.max(np.abs(to.numpy() - o_f)) / np.max(np.abs(o_f)))
            ed = float(np.max(np.abs(td.numpy() - d_f)) / np.max(np.abs(d_f)))
            q_out.put((rank, (wtype, ffn), eo, ed))
    finally:
        dist.destroy_process_group()


def test_tensor_parallel_shards_reproduce_the_unsharded_layer():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    res = [q.get(timeout=10) for _ in range(world * 3)]
    for rank, wtype, eo, ed in res:
        assert eo < 1e-5 and ed < 1e-5, (rank, wtype, eo, ed)
