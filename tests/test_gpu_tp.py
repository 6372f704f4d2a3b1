"""Tensor-parallel plumbing on ONE GPU (the multi-GPU run itself is the driver's): an RCCL communicator of one rank is
created through the C ABI, its all-reduce is stream-ordered and capturable, and a rank-0 shard of a tp_size-2 model decodes
identically whether its (single-rank, hence identity) all-reduce goes through RCCL inside the captured decode graph or
through a host callback launched eagerly."""
import ctypes as C

import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm(gpu):
    L = gpu.lib.get()
    raw = (C.c_char * 128)()
    gpu.lib.check(L.cllm_tp_unique_id(raw), "tp_unique_id")
    c = C.c_void_p()
    gpu.lib.check(L.cllm_tp_init(raw, 0, 1, C.byref(c)), "tp_init")
    yield c
    gpu.lib.check(L.cllm_tp_destroy(c), "tp_destroy")


def test_single_rank_all_reduce_is_identity_and_stream_ordered(gpu, comm):
    L = gpu.lib.get()
    x = np.random.default_rng(0).standard_normal(4096).astype(np.float32)
    d = gpu.Tensor.from_numpy(x.reshape(1, -1))
    gpu.lib.check(L.cllm_tp_all_reduce_f32(comm, None, d.data_ptr(), x.size), "all_reduce")
    assert np.array_equal(d.numpy().ravel(), x)


def test_bad_arguments_are_rejected(gpu):
    L = gpu.lib.get()
    c = C.c_void_p()
    assert L.cllm_tp_init(None, 0, 1, C.byref(c)) != 0
    raw = (C.c_char * 128)()
    assert L.cllm_tp_init(raw, 3, 2, C.byref(c)) != 0          # rank outside the group


def _rank0_shard_model(gpu, cfg, wtype, seed):
    w = gpu.synth.make_model(cfg, wtype, seed=seed)
    hd = cfg["head_dim"]
    QD, F = cfg["n_head"] * hd, cfg["ffn"]
    sh = {}
    for name, (t, arr) in w.items():
        base = name.split(".")[-1]
        if base in ("wq", "wk", "wv", "wgate", "wup"):
            arr = bench.shard_rows(arr, 0, 2)
        elif base == "wo":
            arr = bench.shard_cols(arr, t, QD, 0, 2, gpu)
        elif base == "wdown":
            arr = bench.shard_cols(arr, t, F, 0, 2, gpu)
        sh[name] = (t, arr)
    return sh


def test_rccl_all_reduce_inside_the_decode_graph_matches_the_host_callback(gpu, comm):
    cfg = gpu.synth.config("small", max_len=64, ffn=2048)        # ffn / 256 divisible by the group size
    w = _rank0_shard_model(gpu, cfg, gpu.Q4_K, seed=11)
    prompt = np.random.default_rng(3).integers(0, cfg["vocab"], 7).astype(np.int32)

    a = gpu.Llama(cfg, w, tp_rank=0, tp_size=2)
    a.set_tp_comm(comm)                                   # RCCL, captured in the hipGraph
    la = a.forward(prompt)
    ids_a = a.decode_greedy(int(np.argmax(la)), 24)

    b = gpu.Llama(cfg, w, tp_rank=0, tp_size=2)
    calls = []
    b.set_allreduce(lambda stream, buf, n: calls.append(n))   # identity all-reduce of a one-rank "group", eager launches
    lb = b.forward(prompt)
    ids_b = b.decode_greedy(int(np.argmax(lb)), 24)

    assert np.array_equal(la, lb)
    assert np.array_equal(ids_a, ids_b)
    assert len(calls) == 2 * cfg["n_layer"] * (1 + 24)    # two all-reduces per layer per forward

    # the fused TP step folds "x += all-reduced partial" into the next mat-vec's RMS_NORM prologue; the node-by-node path adds
    # with an ADD node: same bits
    import os
    c = gpu.Llama(cfg, w, tp_rank=0, tp_size=2)
    c.set_allreduce(lambda stream, buf, n: None)
    lc = c.forward(prompt)
    os.environ["CLLM_NO_FUSED"] = "1"
    try:
        ids_c = c.decode_greedy(int(np.argmax(lc)), 24)
    finally:
        del os.environ["CLLM_NO_FUSED"]
    assert np.array_equal(ids_a, ids_c)
    la2 = a.decode_fused_logits(int(ids_a[-1]))
    lc2 = c.forward([int(ids_c[-1])])
    assert np.array_equal(la2, lc2)
    a.close(); b.close(); c.close()


def test_bench_tp_setup_under_torchrun_with_one_rank(gpu):
    """bench.py's multi-GPU set-up (torch first, id broadcast, collective success flag, communicator bound to the runner) walked with
    ONE rank under the driver's launcher -- the only part of the N > 1 path a single-GPU box can execute"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # twice: as on a node (the one-shot all-reduce needs fine-grained IPC memory on every rank, else the decode steps stay on RCCL -- either way said out loud), and with the
    # "all ranks share this GPU" statement that admits coarse-grained memory: there the one-shot path must pass its start-up self-check and carry the steps
    for extra, must in (({}, "decode all-reduces"), ({"CLLM_TP_ONESHOT_SAME_DEVICE": "1"}, "self-check passed")):
        env = dict(os.environ, CLLM_BENCH_TP_SELFTEST="1", **extra)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533",
                            os.path.join(root, "bench.py"), "--gpus", "1", "--model", "small", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-kernels"],
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "tensor parallel over RCCL" in r.stderr and "reports rank 0 of 1" in r.stderr
        assert must in r.stderr, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        res = json.loads(line)
        assert res["n_gpus"] == 1 and res["value"] > 0 and res["metric"] == "decode tokens/s"
        assert res["config"]["rccl_ranks"] == 1 and res["config"]["decode_allreduce"]


def _run_two_ranks(tmp_path, seed, mode, shape="even"):
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / f"tp_{mode}_{shape}.npz")
    env = dict(os.environ, TP_WORKER_MODE=mode, TP_WORKER_CFG=shape, HSA_ENABLE_IPC_MODE_LEGACY="0", CLLM_TP_ONESHOT_SAME_DEVICE="1")      # (both ranks run on the one GPU: the coarse-grained fallback is valid here)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "tp_two_ranks_worker.py"), str(r), "2", str(port), out, str(seed)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    errs = []
    for p in procs:
        try:
            _, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        errs.append(e)
    assert all(p.returncode == 0 for p in procs), [e[-1500:] for e in errs]
    return np.load(out)


def test_one_shot_all_reduce_between_two_processes_equals_the_gloo_all_reduce(gpu, tmp_path):
    """tp_oneshot.hip with two REAL ranks (two processes on this GPU, each other's receive buffer mapped through HIP IPC): every logit of the teacher-forced steps
    and every free-running id equals the run whose all-reduce is a host round trip over gloo (a two-term sum has one order) -- and here the decode steps replay
    from the captured graph with the all-reduce kernels inside."""
    a = _run_two_ranks(tmp_path, 17, "gloo")
    b = _run_two_ranks(tmp_path, 17, "oneshot")
    assert int(b["oneshot_error"]) == 0
    assert np.array_equal(a["logits"].view(np.uint32), b["logits"].view(np.uint32))
    assert np.array_equal(a["ids"], b["ids"])


@pytest.mark.parametrize("shape", ["even", "uneven"])
def test_fused_all_reduce_between_two_processes_equals_the_gloo_all_reduce(gpu, tmp_path, shape):
    """gemv_tp.hip with two REAL ranks (two processes on this GPU, each other's receive buffers mapped through HIP IPC): the single-token steps run WITHOUT any
    all-reduce launch -- o / down send their partial rows as {value, step} granules into both ranks' buffers (EPI 4), the next RMS_NORM mat-vec gathers them in
    rank order (PRO 5) -- eagerly and replayed from the captured graph; every logit of the teacher-forced steps and every free-running id equals the run whose
    all-reduce is a host round trip over gloo (a two-term sum has one order).  "uneven": the Qwen2-style block of the test below (q / k / v biases, NEOX RoPE, a Q8_0
    down_proj of 89 blocks cut 45 + 44: the scatter form of the 32-element formats)."""
    a = _run_two_ranks(tmp_path, 23, "gloo", shape)
    b = _run_two_ranks(tmp_path, 23, "fused", shape)
    assert int(b["fused_error"]) == 0
    assert np.array_equal(a["logits"].view(np.uint32), b["logits"].view(np.uint32))
    assert np.array_equal(a["ids"], b["ids"])
    assert int(b["calls"]) < int(a["calls"])                    # the fused steps made no all-reduce calls (the prompt and the node-by-node steps did)


@pytest.mark.parametrize("shape", ["even", "uneven"])
def test_two_ranks_share_the_gpu_and_all_reduce_over_gloo(gpu, tmp_path, shape):
    """The HIP tensor-parallel path with REAL partial sums: two processes, both on this GPU, each holding one shard of the model; the runner's all-reduce
    is the host callback over gloo (tests/tp_two_ranks_worker.py).  Against the unsharded runner on the same tokens: the sharded o / down outputs are sums of
    two partials in another fp32 order (tier T1 per op), which the activation quantizers downstream may amplify -- teacher-forced logits within 0.25 sigma,
    the argmax equal wherever the unsharded top-1 margin exceeds twice the observed deviation; free-running ids compared the same way.
    shape "uneven": a Qwen2-style block whose Q8_0 down_proj has 89 quant blocks -- rank 0 holds 45 of them (and the matching gate / up rows), rank 1 holds 44
    (cllm_llama_config.ffn_local; BASELINE cfg4 on 8 ranks is 4 x 116 + 4 x 115 of 924)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "tp.npz")
    seed = 17
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "tp_two_ranks_worker.py"), str(r), "2", str(port), out, str(seed)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, TP_WORKER_CFG=shape)) for r in range(2)]
    errs = []
    for p in procs:
        try:
            _, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        errs.append(e)
    assert all(p.returncode == 0 for p in procs), [e[-1500:] for e in errs]
    got = np.load(out)
    sys.path.insert(0, os.path.join(root, "tests"))
    import tp_two_ranks_worker
    os.environ["TP_WORKER_CFG"] = shape
    try:
        cfg = tp_two_ranks_worker.worker_cfg(gpu)
    finally:
        del os.environ["TP_WORKER_CFG"]
    w = gpu.synth.make_model(cfg, gpu.Q4_K, seed=seed)
    ref = gpu.Llama(cfg, w)
    prompt = np.random.default_rng(seed).integers(0, cfg["vocab"], 12).astype(np.int32)
    teacher = np.random.default_rng(seed + 1).integers(0, cfg["vocab"], 10).astype(np.int32)
    want = [ref.forward(prompt)] + [ref.forward([int(t)]) for t in teacher]
    want = np.stack(want)
    assert int(got["calls"]) == 2 * cfg["n_layer"] * (1 + 10 + 8)                 # two all-reduces per layer per forward / step
    sigma = float(np.std(want))
    dev = np.max(np.abs(got["logits"] - want), axis=1)
    assert float(dev.max()) < 0.25 * sigma, (float(dev.max()), sigma)
    top2 = np.sort(want, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2 * dev
    assert np.all(np.argmax(got["logits"], axis=1)[clear] == np.argmax(want, axis=1)[clear])
    # (observed: 0.08-0.10 sigma at EVERY step -- the two-partial sums differ from the unsharded fp32 order in the last bits, the Q8_K activation quantizers of the
    #  next mat-vec turn that into flipped int8 steps, and from the first layer on the two runs sit at the quantization-noise floor: tensor parallelism is a
    #  tolerance-tier path by construction)
    ids_ref = ref.decode_greedy(int(np.argmax(want[-1])), 8)
    print(f"two ranks on one GPU: max|dlogit| {float(dev.max()):.3e} (sigma {sigma:.3f}), median {float(np.median(dev)):.3e}; greedy ids equal: {bool(np.array_equal(ids_ref, got['ids']))}")
    ref.close()
