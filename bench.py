#!/usr/bin/env python3
"""bench.py -- decode tokens/s of the hot path on N MI355X (BASELINE.json metric), one JSON line.

A "step" = one decoded token: the whole forward graph of SURVEY.md 3.3 (embedding gather, n_layer x
[RMSNorm, q/k/v GEMV, RoPE, KV-cache write, attention, o GEMV, residual, RMSNorm, gate/up GEMV, SiLU*up,
down GEMV, residual], final norm, lm_head GEMV) + greedy argmax, on synthetic quantized weights at the
real Llama-3-8B shapes (configs[1]: Q4_K, batch 1).  Weights and KV cache are resident in HBM before
the timed region.

  N == 1 : the whole model on one GPU.
  N  > 1 : tensor-parallel shards (heads / ffn columns) with an all-reduce (RCCL over xGMI) on the
           residual stream after o_proj and down_proj ("scaling": "strong": the same model, N GPUs).

Extra objects on the JSON line (rank 0, N = 1):
  "roofline"     the dominant kernel (the gate|up GEMV): HIP events on the launch stream; "traffic" = HBM bytes per launch from a separate rocprofv3 --pmc FETCH_SIZE
                 pass that bench.py spawns over the same kernel (--no-pmc: null)
  "cpu_baseline" the reference HOST itself (oracle/_ref/ref_chat = chatllm.cpp's graph builder + ggml scheduler + CPU backend, compiled from
                 /root/reference) decoding the same synthetic model end to end on this box's cores: median of 3 runs; "host_cores" = cores of the
                 box, "cores" = threads used (best of an ascending sweep 8 .. all cores with an early stop: the reference's thread pool peaks at 16 on the 256-core box),
                 "build" = x86-64-v3 (the parity oracle) or avx512 (baseline only), whichever is faster
  "other_types"  the north star's other two weight types at the same shapes (Q4_0, Q8_0): tok/s and model-level roofline fraction of the same 20-step measurement
  "dropin"       the SAME unmodified host with every layer on our ggml module (-ngl all): the through-the-boundary number (also "dropin_tok_s")
  "prefill"      BASELINE cfg3: Llama-3-8B shapes, Q4_0, one 4096-token prompt through the runner (median of 3), fraction of the matrix-core peak, in the default
                 (exact-order) mode and in the opt-in fast mode
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
WTYPES = {"q4_k": 12, "q4_0": 2, "q4_1": 3, "q8_0": 8}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def tp_split(n_units, world, rank):
    """whole-unit split, the first n_units % world ranks hold one unit more (== cllm_tp_split, include/chatllm_hip.h): (first, count)"""
    base, extra = divmod(n_units, world)
    return base * rank + min(rank, extra), base + (1 if rank < extra else 0)


def shard_rows(arr, rank, n):
    r = arr.shape[0] // n
    return np.ascontiguousarray(arr[rank * r:(rank + 1) * r])


def shard_cols(arr, type_, K, rank, n, pkg):
    """slice the K (input) dimension of quantized rows in WHOLE blocks; the blocks need not divide evenly (Qwen2-72B's Q8_0 down_proj: 924 blocks over 8 ranks)"""
    bs, blk = pkg.tensor.TYPE_SIZE[type_], pkg.tensor.BLCK[type_]
    nb = K // blk
    first, cnt = tp_split(nb, n, rank)
    a = arr.reshape(arr.shape[0], nb, bs)
    return np.ascontiguousarray(a[:, first:first + cnt, :]).reshape(arr.shape[0], cnt * bs)


def ffn_share(cfg, down_type, rank, world, pkg):
    """this rank's ffn features = the columns of its down_proj blocks: (first feature, count)"""
    blk = pkg.tensor.BLCK[down_type]
    first, cnt = tp_split(cfg["ffn"] // blk, world, rank)
    return first * blk, cnt * blk


def shard_plan(pkg, cfg, wtype, world):
    """shape math of the tensor-parallel shards (no GPU, no weights): per rank the local sizes and the bytes of every sharded tensor; raises where a split is impossible"""
    S = pkg.synth
    H, hd, F = cfg["hidden"], cfg["head_dim"], cfg["ffn"]
    if cfg["n_head"] % world or cfg["n_kv_head"] % world:
        raise ValueError(f"heads {cfg['n_head']} / kv heads {cfg['n_kv_head']} do not divide over {world} ranks")
    dt = S.down_type(cfg, wtype)
    rs = pkg.tensor.row_size
    plan, covered = [], 0
    for r in range(world):
        f0, fl = ffn_share(cfg, dt, r, world, pkg)
        assert f0 == covered and fl > 0 and fl % pkg.tensor.BLCK[dt] == 0 and fl % 8 == 0, (r, f0, fl)
        covered += fl
        nh, nkv = cfg["n_head"] // world, cfg["n_kv_head"] // world
        plan.append({"rank": r, "n_head": nh, "n_kv_head": nkv, "ffn_first": f0, "ffn_local": fl, "down_type": dt, "down_blocks": fl // pkg.tensor.BLCK[dt],
                     "bytes_per_layer": {"wqkv": (nh + 2 * nkv) * hd * rs(wtype, H), "wo": H * rs(wtype, nh * hd), "wgu": 2 * fl * rs(wtype, H), "wdown": H * rs(dt, fl)}})
    assert covered == F
    return plan


def build_model(pkg, cfg, wtype, rank, world):
    """generate (this rank's shard of) the synthetic model tensor by tensor and upload it"""
    S = pkg.synth
    H, hd, F = cfg["hidden"], cfg["head_dim"], cfg["ffn"]
    QD = cfg["n_head"] * hd
    f0, fl = ffn_share(cfg, S.down_type(cfg, wtype), rank, world, pkg) if world > 1 else (0, F)
    m = pkg.Llama(cfg, None, tp_rank=rank, tp_size=world, ffn_local=fl if world > 1 else 0)
    t0 = time.time()
    for name, t, rows, K in S.tensor_list(cfg, wtype):
        a = S.make_tensor_fast(name, t, rows, K)
        base = name.split(".")[-1]
        if world > 1:
            if base in ("wq", "wk", "wv"):
                a = shard_rows(a, rank, world)
            elif base in ("wgate", "wup"):
                a = np.ascontiguousarray(a[f0:f0 + fl])           # the features of this rank's down_proj blocks
            elif base == "wo":
                a = shard_cols(a, t, QD, rank, world, pkg)
            elif base == "wdown":
                a = shard_cols(a, t, F, rank, world, pkg)
        m.set_weight(name, t, a)
    m.set_weight("out_norm", pkg.F32, S.make_norm("out_norm", H))
    for i in range(cfg["n_layer"]):
        p = f"layers.{i}."
        m.set_weight(p + "attn_norm", pkg.F32, S.make_norm(p + "attn_norm", H))
        m.set_weight(p + "ffn_norm", pkg.F32, S.make_norm(p + "ffn_norm", H))
        if cfg.get("qkv_bias"):
            for b, n in (("bq", QD), ("bk", cfg["n_kv_head"] * hd), ("bv", cfg["n_kv_head"] * hd)):
                v = S.make_bias(p + b, n)
                m.set_weight(p + b, pkg.F32, shard_rows(v, rank, world) if world > 1 else v)
    log(f"[rank {rank}] model generated + uploaded in {time.time() - t0:.1f}s")
    return m


def measure_dominant_kernel(pkg, cfg, wtype, iters=64):
    """the decode step's gate/up mat-vec exactly as the runner launches it (RMS_NORM + quantize prologue, 2*ffn x hidden
    weight rows, SiLU(gate)*up epilogue for Q4_K): avg launch duration from HIP events on the launch stream.
    Cycles through enough distinct weight copies to defeat the 256 MiB Infinity Cache."""
    L = pkg.lib.get()
    H, F = cfg["hidden"], cfg["ffn"]
    rows = 2 * F
    rs = pkg.tensor.row_size(wtype, H)
    nbytes = rows * rs
    n_copies = max(2, int(1.5 * 2**30 // nbytes) + 1)
    w0 = pkg.synth.make_tensor_fast("bench.wgu", wtype, rows, H)
    ws = []
    for _ in range(n_copies):
        ws.append(pkg.Tensor.from_numpy(w0, wtype, [H, rows]))
    rng = np.random.default_rng(0)
    x = pkg.Tensor.from_numpy(rng.standard_normal((1, H)).astype(np.float32))
    g = pkg.Tensor.from_numpy((1 + 0.1 * rng.standard_normal((1, H))).astype(np.float32))
    y = pkg.Tensor(pkg.F32, [rows, 1])
    ptrs = (C.c_void_p * n_copies)(*[w.data_ptr().value for w in ws])
    us = C.c_float()
    epi = 1 if F % 8 == 0 else 0
    pkg.lib.check(L.cllm_bench_gemv_fused(None, wtype, ptrs, n_copies, H, rows, 1, x.data_ptr(), g.data_ptr(), cfg["rms_eps"], epi, y.data_ptr(), None,
                                          iters, C.byref(us)), "bench_gemv_fused")
    dur_s = us.value / 1e6
    name = "k_gemv_dec<%d, 1, %d, %d, false, false>" % (wtype, epi, 1 if H <= 4096 else 4)      # (as rocprofv3 prints it: FMT, PRO, EPI, NPRE, MOE, FREE)
    return {"kernel": "%s (gate/up GEMV %dx%d, decode form)" % (name, rows, H), "bytes_per_launch": nbytes, "avg_us": dur_s * 1e6, "gbs": nbytes / dur_s / 1e9}


def kernel_table(pkg, cfg, wtype, n_ctx, iters=48):
    """the launches of one decoder layer + lm_head, each timed in this process with HIP events over back-to-back launches (launch-to-launch time, weight copies cycled
    past the Infinity Cache): [{name, kernel, us, bytes, gbs, frac}] -- frac = algorithmic bytes / time / 8 TB/s.  Attention: 64 launches captured in a hipGraph."""
    L = pkg.lib.get()
    H, hd, F, V = cfg["hidden"], cfg["head_dim"], cfg["ffn"], cfg["vocab"]
    QD, KD = cfg["n_head"] * hd, cfg["n_kv_head"] * hd
    dt = pkg.synth.down_type(cfg, wtype)
    rng = np.random.default_rng(0)
    rows = []

    def gemv(name, t, K, N, pro, epi, resid):
        rs = pkg.tensor.row_size(t, K)
        nbytes = N * rs
        n_copies = max(2, int(1.2 * 2**30 // nbytes) + 1)
        w0 = pkg.synth.make_tensor_fast("kt." + name, t, N, K)
        ws = [pkg.Tensor.from_numpy(w0, t, [K, N]) for _ in range(n_copies)]
        x = pkg.Tensor.from_numpy(rng.standard_normal((1, K)).astype(np.float32))
        g = pkg.Tensor.from_numpy((1 + 0.1 * rng.standard_normal((1, K))).astype(np.float32))
        y = pkg.Tensor(pkg.F32, [N, 1])
        r = pkg.Tensor.from_numpy(rng.standard_normal((1, N)).astype(np.float32)) if resid else None
        ptrs = (C.c_void_p * n_copies)(*[w.data_ptr().value for w in ws])
        us = C.c_float()
        pkg.lib.check(L.cllm_bench_gemv_fused(None, t, ptrs, n_copies, K, N, pro, x.data_ptr(), g.data_ptr(), cfg["rms_eps"], epi, y.data_ptr(),
                                              r.data_ptr() if r is not None else None, iters, C.byref(us)), "bench_gemv_fused " + name)
        rows.append({"name": name, "us": round(us.value, 3), "bytes": nbytes, "gbs": round(nbytes / us.value / 1e3, 1), "frac": round(nbytes / us.value / 1e3 / HBM_PEAK_GBS, 4)})
        del ws

    gemv("qkv (norm + quantize prologue)", wtype, H, QD + 2 * KD, 1, 0, False)
    # attention: RoPE + KV write + scores + soft-max + V.P of one token at n_ctx cached positions
    try:
        ML = max(64, (n_ctx + 63) // 64 * 64)
        st = C.c_void_p()
        pkg.lib.check(L.cllm_stream_create(C.byref(st)), "stream_create")
        qkv = pkg.Tensor.from_numpy(rng.standard_normal((1, QD + 2 * KD)).astype(np.float32))
        kc = pkg.Tensor.from_numpy((rng.standard_normal((ML, KD)) * 0.5).astype(np.float16))
        vc = pkg.Tensor.from_numpy((rng.standard_normal((KD, ML)) * 0.5).astype(np.float16))
        pos = pkg.Tensor.from_numpy(np.array([n_ctx - 1], np.int32))
        cs = pkg.Tensor(pkg.F32, [hd])
        out = pkg.Tensor(pkg.F32, [QD])
        ws = L.cllm_attn_decode_wsize(n_ctx, cfg["n_head"], ML)
        buf = pkg.tensor.Buffer(ws) if ws else None

        def launch():
            return L.cllm_op_rope_kv_attn_decode(st, qkv.data_ptr(), pos.data_ptr(), cs.data_ptr(), cfg["rope_theta"], n_ctx, cfg["n_head"], cfg["n_kv_head"], hd,
                                                 cfg.get("rope_mode", 0), kc.data_ptr(), vc.data_ptr(), ML, out.data_ptr(), buf.ptr if buf else None, buf.nbytes if buf else 0)
        pkg.lib.check(L.cllm_op_rope_table(st, pos.data_ptr(), hd, cfg["rope_theta"], cs.data_ptr()), "rope_table")
        pkg.lib.check(launch(), "attn warm-up")
        pkg.lib.check(L.cllm_stream_sync(st), "sync")
        pkg.lib.check(L.cllm_graph_capture_begin(st), "capture")
        for _ in range(64):
            pkg.lib.check(launch(), "attn capture")
        ge_ = C.c_void_p()
        pkg.lib.check(L.cllm_graph_capture_end(st, C.byref(ge_)), "capture_end")
        e0, e1 = C.c_void_p(), C.c_void_p()
        pkg.lib.check(L.cllm_event_create(C.byref(e0)), "event"); pkg.lib.check(L.cllm_event_create(C.byref(e1)), "event")
        pkg.lib.check(L.cllm_graph_launch(ge_, st), "graph")
        pkg.lib.check(L.cllm_event_record(e0, st), "record")
        for _ in range(8):
            pkg.lib.check(L.cllm_graph_launch(ge_, st), "graph")
        pkg.lib.check(L.cllm_event_record(e1, st), "record")
        pkg.lib.check(L.cllm_event_sync(e1), "event_sync")
        ms = C.c_float()
        pkg.lib.check(L.cllm_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
        us = ms.value * 1e3 / (8 * 64)
        kvb = 2 * n_ctx * KD * 2
        rows.append({"name": "attention (RoPE, KV write, scores, soft-max, V.P; %d cached positions)" % n_ctx, "us": round(us, 3), "bytes": kvb, "gbs": round(kvb / us / 1e3, 1),
                     "frac": round(kvb / us / 1e3 / HBM_PEAK_GBS, 4)})
        L.cllm_graph_destroy(ge_); L.cllm_event_destroy(e0); L.cllm_event_destroy(e1); L.cllm_stream_destroy(st)
    except Exception as e:      # noqa: BLE001
        rows.append({"name": "attention", "error": str(e)})
    gemv("o (quantize prologue, + residual)", wtype, QD, H, 2, 0, True)
    gemv("gate/up (norm + quantize prologue, SiLU * up epilogue)", wtype, H, 2 * F, 1, 1 if F % 8 == 0 else 0, False)
    gemv("down (quantize prologue, + residual)", dt, F, H, 2, 0, True)
    gemv("lm_head (final norm prologue)", wtype, H, V, 1, 0, False)
    layer = [r for r in rows[:5] if "us" in r]
    return {"per_launch": rows, "layer_us": round(sum(r["us"] for r in layer), 2), "layer_bytes": sum(r["bytes"] for r in layer),
            "layer_frac": round(sum(r["bytes"] for r in layer) / sum(r["us"] for r in layer) / 1e3 / HBM_PEAK_GBS, 4) if layer else None,
            "how": "HIP events over back-to-back launches of each kernel in this process (launch-to-launch time), weight copies cycled past the Infinity Cache"}


def ceilings(pkg, with_library_gemm=True):
    """measured ceilings next to the nominal peaks: a pure streaming read (cllm_bench_read_bw: 2 GiB, 16-byte loads) and the library fp16 GEMM
    (torch.matmul = hipBLASLt / rocBLAS at 8192^3, in a subprocess: torch brings its own HIP runtime)"""
    import re
    import subprocess
    out = {"hbm_nominal_TBps": HBM_PEAK_GBS / 1e3, "fp16_mfma_nominal_TFLOPs": 2500.0}
    try:
        gbs = C.c_float()
        pkg.lib.check(pkg.lib.get().cllm_bench_read_bw(None, 2 << 30, 6, C.byref(gbs)), "bench_read_bw")
        out["hbm_read_TBps"] = round(gbs.value / 1e3, 3)
    except Exception as e:      # noqa: BLE001
        out["hbm_read_error"] = str(e)
    if with_library_gemm:
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_ceiling.py")], capture_output=True, text=True, timeout=300)
            m = re.search(r"fp16 \d+\^3: [0-9.]+ ms\s+(\d+) TFLOP/s", r.stdout)
            mb = re.search(r"bf16 \d+\^3: [0-9.]+ ms\s+(\d+) TFLOP/s", r.stdout)
            mi = re.search(r"int8 \d+\^3: [0-9.]+ ms\s+(\d+) TOP/s", r.stdout)
            if m:
                out["fp16_gemm_TFLOPs"] = float(m.group(1))
            if mb:
                out["bf16_gemm_TFLOPs"] = float(mb.group(1))
            if mi:
                out["int8_gemm_TOPs"] = float(mi.group(1))
            if not m:
                out["library_gemm_error"] = (r.stderr or r.stdout)[-200:]
        except Exception as e:      # noqa: BLE001
            out["library_gemm_error"] = str(e)
    return out


def pmc_traffic_live(kernel_substr, timeout=240):
    """HBM bytes per launch of the dominant kernel, measured NOW: a separate rocprofv3 --pmc FETCH_SIZE pass (with --kernel-trace only, as MI355X_MICROARCH.md's HBM section
    prescribes) over a few launches of exactly that kernel (tools/gemv_bench.py --fused, weight copies cycled so that nothing stays in the Infinity Cache); FETCH_SIZE is
    reported in KB and, on gfx950, at 1/2 of the bytes of a wide coalesced stream: bytes = KB * 1024 * 2.  None when rocprofv3 is not there or the pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not found"
    td = tempfile.mkdtemp(dir="/tmp")
    try:
        cmd = ["rocprofv3", "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", td, "--", sys.executable, os.path.join(ROOT, "tools", "gemv_bench.py"),
               "--fused", "--types", "q4_k", "--shapes", "gate_up_silu", "--iters", "8"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        files = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            return None, "no counter output (rc %d): %s" % (r.returncode, r.stderr[-200:])
        vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0])) if row.get("Counter_Name") == "FETCH_SIZE" and kernel_substr in row["Kernel_Name"]]
        if not vals:
            return None, "kernel %s not in the counter output" % kernel_substr
        vals = vals[len(vals) // 4:] or vals              # skip the warm-up launches
        return int(sum(vals) / len(vals) * 1024 * 2), "rocprofv3 --pmc FETCH_SIZE --kernel-trace (separate pass spawned by bench.py, %d launches averaged; KB x 1024 x 2 per MI355X_MICROARCH.md)" % len(vals)
    except Exception as e:      # noqa: BLE001
        return None, "PMC pass failed: %r" % (e,)
    finally:
        shutil.rmtree(td, ignore_errors=True)


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def write_ggmm(model_name, wtype_name, max_len, td):
    import subprocess
    mp = os.path.join(td, f"{model_name}-{wtype_name}.bin")
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_ggmm.py"), "--config", model_name, "--wtype", wtype_name, "--max-len", str(max_len), "--fast",
                         "--out", mp], capture_output=True, text=True)
    if rc.returncode != 0:
        raise RuntimeError("make_ggmm failed: " + rc.stderr[-400:])
    return mp


PROMPT_IDS = [1, 5, 9, 200, 31, 7, 11, 300, 2, 77, 123, 4567, 89, 1000, 2000, 3000]


def run_ref_chat(mp, ngl, threads, n_decode, env=None, timeout=900, ref_dir=None):
    """oracle/_ref/ref_chat MODEL NGL THREADS N_DECODE - IDS...: returns (tokens/s of its own decode timer, stderr)"""
    import re
    import subprocess
    ref = os.path.join(ref_dir or os.path.join(ROOT, "oracle", "_ref"), "ref_chat")
    r = subprocess.run([ref, mp, ngl, str(threads), str(n_decode), "-"] + [str(i) for i in PROMPT_IDS], capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    m = re.search(r"decode: (\d+) tokens in ([0-9.]+) ms", r.stderr)
    if r.returncode != 0 or not m:
        raise RuntimeError("ref_chat failed (rc %d): %s" % (r.returncode, r.stderr[-400:]))
    return int(m.group(1)) * 1e3 / float(m.group(2)), r.stderr


def cpu_host_end_to_end(mp, n_tokens=64, n_probe=16):
    """the reference's own HOST decoding the synthetic model end to end on its CPU backend -- what a user of the reference gets on this box (-n passed explicitly).
    Thread sweep on short probe runs, ASCENDING {8, 16, 32, 64, 128, all cores}, stopped after two counts in a row below 0.7 x the best so far (each probe bounded to 60 s): on
    the 256-core host of the MI355X box the reference's thread pool peaks at 16 threads (43.8 tok/s) and collapses beyond -- 23.6 at 32, 8.0 at 64, 3.0 at 128, no 16-token
    probe within 300 s at all 256 (profiles/r06_cpu_baseline_thread_sweep.txt: the full sweep, which cost the bench 10 minutes).  Then the median of 3 runs of n_tokens decoded
    tokens at the best count.  The parity oracle is the x86-64-v3 build; where oracle/_ref/avx512 exists (the same sources with the AVX-512 / VNNI branches compiled in,
    `make -C oracle ref-avx512`) and runs on this host, it is timed the same way and the FASTER of the two is the reported value (`build` says which)."""
    cores = host_cores()
    cand = sorted({c for c in (8, 16, 32, 64, 128, cores) if 1 <= c <= cores})
    out = {"unit": "tokens/s", "host_cores": cores, "kind": "reference"}
    builds = {"x86-64-v3": None}
    d512 = os.path.join(ROOT, "oracle", "_ref", "avx512")
    if os.path.exists(os.path.join(d512, "ref_chat")):
        builds["avx512"] = d512
    sweeps, best = {}, None
    for bname, bdir in builds.items():
        try:
            probe, top, below = {}, 0.0, 0
            for th in cand:
                try:
                    probe[th] = round(run_ref_chat(mp, "cpu", th, n_probe, ref_dir=bdir, timeout=60)[0], 2)
                except Exception as e:      # noqa: BLE001  (a thread count that times out is a data point, not a failure)
                    probe[th] = None
                    log(f"cpu baseline probe {bname} @ {th} threads: {str(e)[:120]!r}")
                v = probe[th] or 0.0
                top = max(top, v)
                below = below + 1 if v < 0.7 * top else 0
                if below >= 2:
                    break
            sweeps[bname] = probe
            ok = {t: v for t, v in probe.items() if v}
            if not ok:
                continue
            th = max(ok, key=ok.get)
            runs = sorted(round(run_ref_chat(mp, "cpu", th, n_tokens, ref_dir=bdir)[0], 2) for _ in range(3))
            if best is None or runs[1] > best["value"]:
                best = {"value": runs[1], "cores": th, "runs": runs, "build": bname}
        except Exception as e:      # noqa: BLE001  (e.g. SIGILL: the host lacks an extension the avx512 build assumes)
            sweeps[bname] = {"error": str(e)[-200:]}
    if best is None:
        raise RuntimeError("no CPU run completed: %r" % (sweeps,))
    out.update(best)
    out["thread_sweep_tok_s"] = sweeps
    out["sample"] = ("oracle/_ref/ref_chat (chatllm.cpp's host + ggml CPU backend, built from /root/reference) decoding %d tokens after a 16-token prompt, same synthetic model as a GGMM "
                     "file; ascending thread sweep over %s on %d-token probes per build (stopped once two counts in a row fall below 0.7 x the best), then the median of 3 runs at the "
                     "best count; builds timed: %s, reported: %s at %d threads" % (n_tokens, cand, n_probe, list(builds), best["build"], best["cores"]))
    return out


def dropin_through_the_boundary(mp, n_decode=144):
    """the unmodified reference host with every layer on our ggml module (libggml-hip.so -> the C ABI): decode tokens/s by the host's own timer
    (the first 16 steps are left out: first launches, captures), calls per token from the module's statistics"""
    import re
    tok_s, err = run_ref_chat(mp, "all", 16, n_decode, env={"CLLM_HIP_STATS": "1"})
    calls = [int(x) for x in re.findall(r"graph_compute: \d+ nodes -> (\d+) calls", err)]
    per_graph = re.findall(r"per graph over the last 64: (.*)", err)
    return {"tok_s": tok_s, "ms_per_token": 1e3 / tok_s, "calls_per_token": calls[-1] if calls else None, "n_decode": n_decode,
            "host": "oracle/_ref/ref_chat -ngl all (unmodified chatllm.cpp host, ggml scheduler; module = chatllm.cpp_amd/host/ggml-hip.cpp)",
            "breakdown_us": per_graph[-1] if per_graph else None}


def parity_full_depth(mp, vocab, n_decode=16, cpu_threads=32):
    """the SAME GGMM file through the reference host twice -- its own CPU backend, then every layer on the module -- free-running greedy (each run feeds its own arg-max
    back), the 16-token prompt + n_decode tokens; compares the ids and every logit word (src/models.cpp:1399-1424 walks ALL layers: so does this).  The logits go through
    files in /tmp (ref_chat's own dump: float32, vocab per step)."""
    import subprocess
    import tempfile
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_chat")
    out = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        runs = {}
        for tag, ngl, th in (("cpu", "cpu", min(cpu_threads, host_cores())), ("module", "all", 16)):
            lp = os.path.join(td, tag + ".logits")
            t0 = time.time()
            r = subprocess.run([ref, mp, ngl, str(th), str(n_decode), lp] + [str(i) for i in PROMPT_IDS], capture_output=True, text=True, timeout=3000)
            if r.returncode != 0:
                raise RuntimeError(f"ref_chat ({tag}) failed: " + r.stderr[-400:])
            runs[tag] = ([int(t) for t in r.stdout.split()], np.fromfile(lp, np.float32).reshape(-1, vocab))
            out[tag + "_wall_s"] = round(time.time() - t0, 1)
        (ids_c, lg_c), (ids_g, lg_g) = runs["cpu"], runs["module"]
        n = min(len(lg_c), len(lg_g))
        diff = (lg_c[:n].view(np.uint32) != lg_g[:n].view(np.uint32))
        steps = np.nonzero(diff.any(axis=1))[0]
        out.update({"ids_equal": ids_c == ids_g, "n_ids": len(ids_c), "logit_steps_compared": int(n), "logit_words_differing": int(diff.sum()),
                    "first_differing_step": int(steps[0]) if len(steps) else None, "max_abs_dlogit": float(np.max(np.abs(lg_c[:n] - lg_g[:n]))),
                    "cpu_threads": min(cpu_threads, host_cores())})
    return out


def prefill_cfg3(pkg, model_name, n_prompt=4096, reps=3):
    """BASELINE cfg3: Q4_0 weights, one n_prompt-token prompt through the runner (cllm_llama_forward): median wall time and the algorithmic FLOPs of SURVEY 8d, in the default
    mode (exact: the reference's accumulation order on the K = 4 / f32 matrix-core instructions, bit-identical to the CPU for every prompt length) and in the opt-in
    modes (CLLM_PREFILL=fast: int8-MFMA GEMM + flash attention, tolerance tier; CLLM_PREFILL=f16: dequant -> fp16 MFMA GEMM, a different computation)"""
    cfg = pkg.synth.config(model_name, max_len=(n_prompt + 63) // 64 * 64)
    m = build_model(pkg, cfg, WTYPES["q4_0"], 0, 1)
    prompt = np.random.default_rng(1234).integers(0, cfg["vocab"], n_prompt).astype(np.int32)
    H, hd, F, L = cfg["hidden"], cfg["head_dim"], cfg["ffn"], cfg["n_layer"]
    QD, KD = cfg["n_head"] * hd, cfg["n_kv_head"] * hd
    flops = 2.0 * L * (H * (QD + 2 * KD) + QD * H + 3 * H * F) * n_prompt + 2.0 * 2 * n_prompt * n_prompt * hd * cfg["n_head"] * L
    lib = pkg.lib.get()
    out = {}
    default_mode = lib.cllm_get_prefill_mode()
    for name, mode in (("exact", 1), ("fast", 0), ("f16", 0)):
        pkg.lib.check(lib.cllm_set_prefill_mode(mode), "set_prefill_mode")
        lib.cllm_debug_set_prefill_f16(1 if name == "f16" else 0)          # f16: quantized weights x fp16 activations as a dense fp16 MFMA GEMM (dense_f16.hip), the north star's path B
        try:
            m.forward(prompt, n_past=0)
            pkg.ops.sync()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                m.forward(prompt, n_past=0)
                pkg.ops.sync()
                ts.append(time.perf_counter() - t0)
        finally:
            lib.cllm_set_prefill_mode(default_mode)
            lib.cllm_debug_set_prefill_f16(0)
        dt = sorted(ts)[len(ts) // 2]
        out[name] = {"ms": dt * 1e3, "tok_s": n_prompt / dt, "algorithmic_tflops": flops / dt / 1e12, "frac_of_f16_mfma_peak_2.5e15": flops / dt / 2.5e15, "frac_of_int8_mfma_peak_5e15": flops / dt / 5.0e15}
    m.close()
    d = out["exact" if default_mode == 1 else "fast"]
    res = {"mode": "exact" if default_mode == 1 else "fast", "ms": d["ms"], "tok_s": d["tok_s"], "n_prompt": n_prompt, "wtype": "q4_0", "algorithmic_tflops": d["algorithmic_tflops"],
           "mfma_frac": d["frac_of_f16_mfma_peak_2.5e15"],
           "mfma_peak": "2.5e15 dense fp16 (exact mode: block sums on v_mfma_f32_16x16x4_4b_f16, attention as fmaf chains on v_mfma_f32_16x16x4_f32 -- both legacy-rate instructions; "
                        "its fp32 fold chains run on the VALU, which bounds it: profiles/r03_prefill_modes_gemm.txt)",
           "modes": out}
    return res


def other_type_decode(pkg, cfg, wt, prompt, steps=20, warmup=5, free_order=False):
    """the headline measurement for another weight type: same shapes, same prompt, `steps` greedy tokens after `warmup`; model-level fraction of the 8 TB/s roofline.
    free_order: the opt-in tolerance tier of the 32-weight block formats (gemv_free32.hip) instead of the exact order -- priced next to it, never the default"""
    L = pkg.lib.get()
    L.cllm_set_decode_free_order(1 if free_order else 0)
    m = build_model(pkg, cfg, wt, 0, 1)
    try:
        tok = int(np.argmax(m.forward(prompt, n_past=0)))
        tok = int(m.decode_greedy(tok, warmup)[-1])
        pkg.ops.sync()
        t0 = time.perf_counter()
        out = m.decode_greedy(tok, steps)
        pkg.ops.sync()
        dt = time.perf_counter() - t0
        b = pkg.synth.weight_bytes_per_token(cfg, wt) + pkg.synth.kv_bytes_per_token(cfg, len(prompt) + warmup + steps // 2)
        return {"value": steps / dt, "unit": "tokens/s", "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "algorithmic_bytes_per_token": b,
                "model_hbm_frac": b * steps / dt / (HBM_PEAK_GBS * 1e9), "greedy_tail": [int(t) for t in out[-4:]],
                "contract": ("FREE fp32 fold order (opt-in CLLM_DECODE_FREE_ORDER=1; integer block sums exact): a tolerance tier, NOT the default -- tests/test_gpu_llama.py says what it keeps"
                             if free_order else "exact order (every logit bit-identical to the reference's x86-64-v3 CPU build): the default")}
    finally:
        m.close()
        L.cllm_set_decode_free_order(0)


def layer_split(pkg, cfg, iters=16):
    """where a decoder layer's time goes, from the in-kernel stamps of the four mat-vec launches (s_memrealtime, 100 MHz; thread 0 of every workgroup): per launch
    prologue (entry -> the activation row is in LDS and the barrier has opened), stream (barrier -> the workgroup's last row), boundary (launch-to-launch time minus the
    in-kernel span: dispatch, ramp, drain).  Medians over the workgroups of the last of `iters` launches; the stamps cost ~0.3 us per launch."""
    L = pkg.lib.get()
    lib = C.CDLL(pkg.lib.SO_PATH)
    lib.cllm_debug_set_mmvq_ts.argtypes = [C.c_void_p]
    H, hd, F = cfg["hidden"], cfg["head_dim"], cfg["ffn"]
    QD, KD = cfg["n_head"] * hd, cfg["n_kv_head"] * hd
    rng = np.random.default_rng(0)
    ts = pkg.tensor.Buffer(256 * 8 * 8)
    out = {}
    tot = {"prologue_us": 0.0, "stream_us": 0.0, "boundary_us": 0.0}
    for name, K, N, pro, epi, resid in (("qkv", H, QD + 2 * KD, 1, 0, False), ("o", QD, H, 2, 0, True), ("gate_up", H, 2 * F, 1, 1, False), ("down", F, H, 2, 0, True)):
        nbytes = N * pkg.tensor.row_size(12, K)
        n_copies = max(2, int(1.2 * 2**30 // nbytes) + 1)
        w0 = pkg.synth.make_tensor_fast("ls." + name, 12, N, K)
        ws = [pkg.Tensor.from_numpy(w0, 12, [K, N]) for _ in range(n_copies)]
        x = pkg.Tensor.from_numpy(rng.standard_normal((1, K)).astype(np.float32))
        g = pkg.Tensor.from_numpy((1 + 0.1 * rng.standard_normal((1, K))).astype(np.float32))
        y = pkg.Tensor(pkg.F32, [N, 1])
        r = pkg.Tensor.from_numpy(rng.standard_normal((1, N)).astype(np.float32))
        ptrs = (C.c_void_p * n_copies)(*[w.data_ptr().value for w in ws])
        us = C.c_float()
        L.cllm_memset(ts.ptr, 0, 256 * 64, None)
        lib.cllm_debug_set_mmvq_ts(ts.ptr)
        try:
            pkg.lib.check(L.cllm_bench_gemv_fused(None, 12, ptrs, n_copies, K, N, pro, x.data_ptr(), g.data_ptr(), cfg["rms_eps"], epi, y.data_ptr(),
                                                  r.data_ptr() if resid else None, iters, C.byref(us)), "bench_gemv_fused " + name)
        finally:
            lib.cllm_debug_set_mmvq_ts(None)
        host = np.zeros(256 * 8, dtype=np.uint64)
        pkg.lib.check(L.cllm_memcpy_d2h(host.ctypes.data_as(C.c_void_p), ts.ptr, host.nbytes, None), "d2h")
        L.cllm_stream_sync(None)
        st = host.reshape(256, 8)[:, :6].astype(np.int64)
        st = st[st[:, 5] > 0]
        t0 = st[:, 0].min()
        pro_us = float(np.median(st[:, 3] - st[:, 0])) / 100.0           # per workgroup: its own entry -> its barrier
        end_us = float(np.median(st[:, 5] - st[:, 0])) / 100.0           # per workgroup: its own entry -> its last row (the launch-to-launch time also holds the dispatch skew and the drain)
        row = {"launch_us": round(us.value, 2), "prologue_us": round(pro_us, 2), "stream_us": round(end_us - pro_us, 2), "boundary_us": round(us.value - end_us, 2),
               "stream_gbs": round(nbytes / (end_us - pro_us) / 1e3, 1)}
        out[name] = row
        for k in tot:
            tot[k] += row[k]
        del ws
    out["per_layer_mat_vecs"] = {k: round(v, 2) for k, v in tot.items()}
    out["note"] = "attention (one launch, latency-bound) is in `kernels`; stream_gbs counts the launch's whole matrix against the time after the barrier (the first step was requested at entry)"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512, help="decode steps in the timed region (BASELINE cfg2: 16-token prompt, 512 decoded tokens)")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--wtype", default="q4_k", choices=sorted(WTYPES))
    ap.add_argument("--n-prompt", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the host runs (cpu_baseline, dropin) and the prefill leg")
    ap.add_argument("--no-prefill", action="store_true")
    ap.add_argument("--no-other-types", action="store_true", help="skip the Q4_0 / Q8_0 decode legs (other_types)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc pass that measures the dominant kernel's HBM traffic (roofline.traffic = null)")
    ap.add_argument("--no-kernels", action="store_true", help="skip the per-launch kernel table and the measured ceilings")
    ap.add_argument("--dropin-cfg5", action="store_true", help="also run BASELINE cfg5 (Mixtral-8x7B shapes, Q4_K, 26 GB GGMM file in /tmp) through the unmodified reference host on the module")
    ap.add_argument("--dropin-cfg4", action="store_true", help="also run BASELINE cfg4 on ONE GPU (Qwen2-72B shapes, 50 GB GGMM file in /tmp) through the unmodified reference host on the module")
    ap.add_argument("--no-full-depth-parity", action="store_true", help="with --dropin-cfg4 / --dropin-cfg5: skip the CPU-host-vs-module comparison of ids and logit words over ALL layers")
    ap.add_argument("--dry-run-shards", action="store_true", help="shape math of the N-rank tensor-parallel shards only (no GPU, no weights): one JSON object, then exit")
    ap.add_argument("--no-graph", action="store_true", help="launch the fused decode kernels eagerly (for rocprofv3 kernel traces)")
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.dry_run_shards:
        pkg = ge.load_package()
        n = max(world, args.gpus)
        cfg = pkg.synth.config(args.model, max_len=1024)
        plan = shard_plan(pkg, cfg, WTYPES[args.wtype], n)
        if rank == 0:
            print(json.dumps({"model": args.model, "wtype": args.wtype, "ranks": n, "ffn": cfg["ffn"], "shards": plan}), flush=True)
        return
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    dist = None
    rccl_ranks = None
    allreduce_kind = None
    tp_setup = world > 1 or os.environ.get("CLLM_BENCH_TP_SELFTEST") == "1"          # (self-test: walk the communicator set-up with one rank)
    json_fd = None
    if tp_setup:
        # RCCL (torch's and /opt/rocm's) prints its version banner with printf -- C stdio, flushed at exit, i.e. BEHIND the JSON line on a redirected stdout.  The contract is ONE
        # line on stdout: the process's fd 1 becomes stderr from here on, the JSON line goes to a duplicate of the real stdout.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        # torch FIRST: it bundles its own HIP runtime; initialised after libchatllm_hip.so has loaded /opt/rocm's, it finds "No HIP GPUs"
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    pkg = ge.load_package()
    pkg.lib.require_gpu()
    pkg.lib.check(pkg.lib.get().cllm_set_device(local), "set_device")

    wtype = WTYPES[args.wtype]
    cfg2_steps = 512                       # BASELINE cfg2: 16-token prompt, 512 decoded tokens -- measured next to the requested --steps when they differ
    max_len = (args.n_prompt + max(args.warmup + args.steps, 16 + cfg2_steps) + 8 + 63) // 64 * 64
    cfg = pkg.synth.config(args.model, max_len=max_len)
    m = build_model(pkg, cfg, wtype, rank, world)
    if args.no_graph:
        m.use_graph(False)

    if tp_setup:
        import torch
        L = pkg.lib.get()
        # native path: RCCL communicator owned by the C library; the all-reduces are stream-ordered launches inside the decode graph.
        # Every decision below is taken by ALL ranks together (a rank that fell back alone would leave the others in a collective).
        native = False
        idbuf = torch.zeros(129, dtype=torch.uint8, device=f"cuda:{local}")          # 128 id bytes + "valid"
        if rank == 0:
            try:
                raw = (C.c_char * 128)()
                pkg.lib.check(L.cllm_tp_unique_id(raw), "tp_unique_id")
                idbuf.copy_(torch.frombuffer(bytearray(raw.raw) + bytearray([1]), dtype=torch.uint8))
            except Exception as e:                    # noqa: BLE001
                log(f"[rank 0] cllm_tp_unique_id failed: {e}")
        dist.broadcast(idbuf, 0)
        host_id = bytes(idbuf.cpu().numpy().tobytes())
        if host_id[128] == 1:
            okflag = torch.ones(1, dtype=torch.int32, device=f"cuda:{local}")
            comm = C.c_void_p()
            try:
                pkg.lib.check(L.cllm_tp_init((C.c_char * 128).from_buffer_copy(host_id[:128]), rank, world, C.byref(comm)), "tp_init")
            except Exception as e:                    # noqa: BLE001
                log(f"[rank {rank}] cllm_tp_init failed: {e}")
                okflag.zero_()
            dist.all_reduce(okflag, op=dist.ReduceOp.MIN)
            native = int(okflag.item()) == 1
            if native:
                m.set_tp_comm(comm)
                nr, ur = C.c_int(), C.c_int()
                pkg.lib.check(L.cllm_tp_comm_info(comm, C.byref(nr), C.byref(ur)), "tp_comm_info")
                if nr.value != world or ur.value != rank:
                    raise RuntimeError(f"RCCL reports rank {ur.value} of {nr.value}, the launcher said rank {rank} of {world}")
                rccl_ranks = nr.value
                allreduce_kind = "RCCL (ncclAllReduce inside the decode graph)"
                log(f"[rank {rank}] tensor parallel over RCCL: the communicator reports rank {ur.value} of {nr.value} (all-reduce inside the decode graph)")
                if os.environ.get("CLLM_TP_ONESHOT", "1") != "0":
                    # the decode-sized all-reduces ([hidden] fp32: latency-bound, the wrong regime for a ring) as ONE kernel launch each -- every rank writes its partial vector
                    # into every peer's IPC-mapped receive buffer and sums the slots in rank order (tp_oneshot.hip); prompt-sized messages stay on RCCL.  Taken only if it
                    # can be set up on EVERY rank (fine-grained IPC memory; no coarse-grained fallback across GPUs) AND reproduces a known rank-order sum in a self-check of
                    # six all-reduces right here; otherwise the decode steps keep RCCL.  All ranks decide together.  CLLM_TP_ONESHOT=0: RCCL only.
                    ok1 = torch.ones(1, dtype=torch.int32, device=f"cuda:{local}")
                    osh = C.c_void_p()
                    mine = (C.c_char * 64)()
                    try:
                        pkg.lib.check(L.cllm_tp_oneshot_create(rank, world, cfg["hidden"] * 8, C.byref(osh), mine), "tp_oneshot_create")
                    except Exception as e:                # noqa: BLE001
                        log(f"[rank {rank}] cllm_tp_oneshot_create failed: {e}")
                        ok1.zero_()
                        osh = C.c_void_p()
                    dist.all_reduce(ok1, op=dist.ReduceOp.MIN)                  # (nobody opens handles a peer could not create)
                    if int(ok1.item()) == 1:
                        gathered = [None] * world
                        dist.all_gather_object(gathered, bytes(mine.raw))
                        try:
                            pkg.lib.check(L.cllm_tp_oneshot_connect(osh, b"".join(gathered)), "tp_oneshot_connect")
                        except Exception as e:            # noqa: BLE001
                            log(f"[rank {rank}] cllm_tp_oneshot_connect failed: {e}")
                            ok1.zero_()
                        dist.all_reduce(ok1, op=dist.ReduceOp.MIN)
                    if int(ok1.item()) == 1:
                        # self-check: six all-reduces of known vectors (both slot parities, the sequence counter), the rank-order fp32 sum must come back bit for bit
                        n = cfg["hidden"]
                        xs = [np.random.default_rng(1000 + r).standard_normal(n).astype(np.float32) for r in range(world)]
                        try:
                            for it in range(6):
                                f = np.float32(it + 1)
                                want = xs[0] * f
                                for r in range(1, world):
                                    want = want + xs[r] * f
                                buf = pkg.Tensor.from_numpy((xs[rank] * f).reshape(1, n))
                                pkg.lib.check(L.cllm_tp_oneshot_all_reduce_f32(osh, None, buf.data_ptr(), n), "tp_oneshot_all_reduce")
                                pkg.ops.sync()
                                got = buf.numpy().reshape(n)
                                if L.cllm_tp_oneshot_error(osh) or not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                                    log(f"[rank {rank}] one-shot all-reduce self-check failed at call {it} (max |d| {float(np.max(np.abs(got - want))):.3g})")
                                    ok1.zero_()
                        except Exception as e:            # noqa: BLE001
                            log(f"[rank {rank}] one-shot all-reduce self-check raised: {e}")
                            ok1.zero_()
                        dist.all_reduce(ok1, op=dist.ReduceOp.MIN)
                    if int(ok1.item()) == 1:
                        m.set_tp_oneshot(osh)
                        allreduce_kind = "one-shot direct-write kernel (fine-grained IPC)" if L.cllm_tp_oneshot_fine_grained(osh) == 1 else "one-shot direct-write kernel (coarse-grained: ranks share one GPU)"
                        log(f"[rank {rank}] decode all-reduces through the {allreduce_kind}; self-check passed")
                        if os.environ.get("CLLM_TP_FUSED", "1") != "0":
                            # NO all-reduce launch in the decode steps: o / down send their partial rows as granules into every rank's buffer, the next RMS_NORM mat-vec gathers
                            # them in rank order (gemv_tp.hip).  Taken only if every rank can set it up AND four greedy steps from the same state give the ids and the logit
                            # bits of the one-shot path (same rank-order sums) on every rank; otherwise the one-shot kernel stays.  CLLM_TP_FUSED=0: off.
                            ok2 = torch.ones(1, dtype=torch.int32, device=f"cuda:{local}")
                            fus = C.c_void_p()
                            mine2 = (C.c_char * 64)()
                            try:
                                pkg.lib.check(L.cllm_tp_fused_create(rank, world, 2 * cfg["n_layer"], cfg["hidden"], C.byref(fus), mine2), "tp_fused_create")
                            except Exception as e:            # noqa: BLE001
                                log(f"[rank {rank}] cllm_tp_fused_create failed: {e}")
                                ok2.zero_()
                                fus = C.c_void_p()
                            dist.all_reduce(ok2, op=dist.ReduceOp.MIN)
                            if int(ok2.item()) == 1:
                                g2 = [None] * world
                                dist.all_gather_object(g2, bytes(mine2.raw))
                                try:
                                    pkg.lib.check(L.cllm_tp_fused_connect(fus, b"".join(g2)), "tp_fused_connect")
                                except Exception as e:        # noqa: BLE001
                                    log(f"[rank {rank}] cllm_tp_fused_connect failed: {e}")
                                    ok2.zero_()
                                dist.all_reduce(ok2, op=dist.ReduceOp.MIN)
                            if int(ok2.item()) == 1:
                                try:
                                    probe = np.random.default_rng(4321).integers(0, cfg["vocab"], 8).astype(np.int32)
                                    t0_ = int(np.argmax(m.forward(probe, n_past=0)))
                                    ids_a = m.decode_greedy(t0_, 4)
                                    lg_a = m.debug_read("logits", cfg["vocab"])
                                    m.forward(probe, n_past=0)
                                    m.set_tp_fused(fus)
                                    ids_b = m.decode_greedy(t0_, 4)
                                    lg_b = m.debug_read("logits", cfg["vocab"])
                                    if L.cllm_tp_fused_error(fus) or not np.array_equal(ids_a, ids_b) or not np.array_equal(lg_a.view(np.uint32), lg_b.view(np.uint32)):
                                        log(f"[rank {rank}] fused all-reduce self-check failed (ids {ids_a.tolist()} vs {ids_b.tolist()})")
                                        ok2.zero_()
                                except Exception as e:        # noqa: BLE001
                                    log(f"[rank {rank}] fused all-reduce self-check raised: {e}")
                                    ok2.zero_()
                                dist.all_reduce(ok2, op=dist.ReduceOp.MIN)
                                if int(ok2.item()) == 1:
                                    allreduce_kind = "fused into the mat-vecs (granules scattered by o / down, gathered by the next RMS_NORM launch; no all-reduce launch)"
                                    log(f"[rank {rank}] decode all-reduces {allreduce_kind}; self-check against the one-shot path passed")
                                else:
                                    m.set_tp_fused(None)
                                    log(f"[rank {rank}] fused all-reduce not available on every rank: the decode all-reduces stay on the one-shot kernel")
                    else:
                        log(f"[rank {rank}] one-shot all-reduce not available on every rank: the decode all-reduces stay on RCCL")
        if not native and os.environ.get("CLLM_BENCH_TORCH_ALLREDUCE") != "1":
            # the fallback (torch.distributed.all_reduce behind a stream synchronize per collective, decode steps launched eagerly) measures host round trips, not the
            # path this bench is about: it is an explicit debugging mode, never a silent substitute
            raise RuntimeError("bench.py --gpus N: the RCCL communicator could not be created on every rank (see the messages above); "
                               "CLLM_BENCH_TORCH_ALLREDUCE=1 runs the synchronising torch.distributed callback instead (not a valid scaling measurement)")
        if not native:
            log(f"[rank {rank}] native RCCL path unavailable; CLLM_BENCH_TORCH_ALLREDUCE=1: using the torch.distributed callback (NOT a valid scaling measurement)")

            class _Arr:                       # wrap the raw device pointer for torch (zero copy)
                def __init__(self, ptr, n):
                    self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}

            def allreduce(stream, buf, n):
                pkg.ops.sync()                        # the runner's kernels are on its own stream: order them before torch's collective ...
                dist.all_reduce(torch.as_tensor(_Arr(buf, n), device=f"cuda:{local}"))
                torch.cuda.current_stream().synchronize()          # ... and the collective before the runner continues
            m.set_allreduce(allreduce)

    def sync_all():
        if dist is not None:
            dist.barrier()
        pkg.ops.sync()

    prompt = np.random.default_rng(1234).integers(0, cfg["vocab"], args.n_prompt).astype(np.int32)
    logits = m.forward(prompt, n_past=0)          # (the tensor-parallel self-checks above may have moved the position)
    tok = int(np.argmax(logits))
    if args.warmup > 0:
        tok = int(m.decode_greedy(tok, args.warmup)[-1])
    sync_all()
    t0 = time.perf_counter()
    out = m.decode_greedy(tok, args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        tmax = torch.tensor([dt], device=f"cuda:{local}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    res = {
        "metric": "decode tokens/s", "value": args.steps / dt, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        # the same model on N GPUs (tensor-parallel shards): total work is fixed as N grows
        "scaling": "strong", "vs_baseline": None,
        "dtype": "int8xint4 dot / f32 accumulate (Q8_K x Q4_K)" if wtype == 12 else "int8 dot / f32 accumulate",
        "data": "synthetic",
        "config": {"workload": f"{args.model} shapes, {args.wtype.upper()} weights, single-token decode, batch 1, {args.n_prompt}-token prompt, F16 KV cache",
                   "parallelism": f"tp{world}" if world > 1 else "single GPU", "n_ctx_end": args.n_prompt + args.warmup + args.steps,
                   "rccl_ranks": rccl_ranks, "decode_allreduce": allreduce_kind},
    }
    if rank == 0:
        n_ctx = args.n_prompt + args.warmup + args.steps // 2
        bytes_tok = pkg.synth.weight_bytes_per_token(cfg, wtype) + pkg.synth.kv_bytes_per_token(cfg, n_ctx)
        res["model_hbm_frac"] = bytes_tok * res["value"] / (HBM_PEAK_GBS * 1e9) / world
        res["algorithmic_bytes_per_token"] = bytes_tok
        res["greedy_tail"] = [int(t) for t in out[-4:]]
        if world == 1:
            if args.steps != cfg2_steps and args.model == "llama3-8b":
                # the same metric at BASELINE cfg2's own length (16-token prompt, 512 decoded tokens: n_ctx 32 -> 544), whatever --steps the caller timed
                try:
                    lg2 = m.forward(prompt, n_past=0)
                    t2 = int(m.decode_greedy(int(np.argmax(lg2)), 16)[-1])
                    pkg.ops.sync()
                    t0 = time.perf_counter()
                    m.decode_greedy(t2, cfg2_steps)
                    pkg.ops.sync()
                    dt2 = time.perf_counter() - t0
                    b2 = pkg.synth.weight_bytes_per_token(cfg, wtype) + pkg.synth.kv_bytes_per_token(cfg, args.n_prompt + 16 + cfg2_steps // 2)
                    res["decode_512"] = {"value": cfg2_steps / dt2, "unit": "tokens/s", "steps": cfg2_steps, "warmup": 16, "ms_per_step": dt2 / cfg2_steps * 1e3,
                                         "model_frac": b2 * (cfg2_steps / dt2) / (HBM_PEAK_GBS * 1e9)}
                except Exception as e:      # noqa: BLE001
                    res["decode_512"] = {"error": str(e)}
            try:
                k = measure_dominant_kernel(pkg, cfg, wtype)
                traffic, tsrc = (None, "skipped (--no-pmc)") if args.no_pmc or wtype != 12 else pmc_traffic_live("k_gemv_dec")
                res["roofline"] = {"bound": "hbm", "achieved": k["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["gbs"] / HBM_PEAK_GBS,
                                   "traffic": traffic, "traffic_source": tsrc, "kernel": k["kernel"], "avg_us": k["avg_us"], "bytes_per_launch": k["bytes_per_launch"],
                                   # the whole decode step against the same peak: algorithmic bytes per token x tokens/s / 8 TB/s (what the north star's 0.70 is about)
                                   "model_frac": res["model_hbm_frac"]}
            except Exception as e:      # the throughput number stands on its own
                res["roofline"] = {"error": str(e)}
            if not args.no_kernels and wtype == 12:
                try:
                    res["layer_split"] = layer_split(pkg, cfg)
                except Exception as e:      # noqa: BLE001
                    res["layer_split"] = {"error": str(e)}
            if not args.no_kernels:
                try:
                    res["kernels"] = kernel_table(pkg, cfg, wtype, args.n_prompt + args.warmup + args.steps // 2)
                except Exception as e:      # noqa: BLE001
                    res["kernels"] = {"error": str(e)}
                res["ceilings"] = ceilings(pkg, with_library_gemm=not args.no_cpu_baseline)
            res["host_cores"] = host_cores()
            if not args.no_cpu_baseline:
                import tempfile
                m.close()                                    # the host runs below load their own copy of the model
                m = None
                with tempfile.TemporaryDirectory(dir="/tmp") as td:
                    mp = None
                    try:
                        mp = write_ggmm(args.model, args.wtype, 512, td)
                    except Exception as e:      # the baselines are reports, never a reason to lose the bench line
                        log(f"GGMM file: {e!r}")
                    try:
                        res["cpu_baseline"] = cpu_host_end_to_end(mp) if mp else {"error": "no GGMM file"}
                    except Exception as e:
                        res["cpu_baseline"] = {"error": str(e)}
                    try:
                        res["dropin"] = dropin_through_the_boundary(mp) if mp else {"error": "no GGMM file"}
                        res["dropin_tok_s"] = res["dropin"].get("tok_s")          # the through-the-boundary number (unmodified reference host on the module), first class
                    except Exception as e:
                        res["dropin"] = {"error": str(e)}
                for flag, key, arch, cname, ndec in ((args.dropin_cfg5, "dropin_cfg5", "mixtral", "mixtral-8x7b", 272), (args.dropin_cfg4, "dropin_cfg4_one_gpu", "qwen2", "qwen2-72b", 144)):
                    if not flag:
                        continue
                    try:
                        import subprocess
                        mp5 = f"/tmp/{cname}-q4_k.bin"
                        if not os.path.exists(mp5):
                            rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_ggmm.py"), "--arch", arch, "--config", cname, "--wtype", "q4_k", "--max-len", "512",
                                                 "--fast", "--out", mp5], capture_output=True, text=True)
                            if rc.returncode != 0:
                                raise RuntimeError("make_ggmm failed: " + rc.stderr[-300:])
                        res[key] = dropin_through_the_boundary(mp5, n_decode=ndec)
                        res[key]["model"] = cname + " shapes, Q4_K (down_proj per the reference's fallback rule), 16-token prompt"
                        if not args.no_full_depth_parity:
                            try:                      # ALL layers of the config, CPU host vs module, ids and logit words
                                res[key]["parity"] = parity_full_depth(mp5, pkg.synth.config(cname, max_len=512)["vocab"], n_decode=16)
                            except Exception as e:    # noqa: BLE001
                                res[key]["parity"] = {"error": str(e)}
                    except Exception as e:      # noqa: BLE001
                        res[key] = {"error": str(e)}
                if args.model == "llama3-8b" and wtype == WTYPES["q4_k"] and not args.no_other_types:
                    # the north star's other two weight types at the same shapes (20 steps after a 5-step warm-up, the driver's own numbers): the exact-order kernels of
                    # gemv_rows32.hip / gemv_team32.hip (8 per-AVX-lane sums and 8 chain steps per 32 weights: DESIGN.md section 6, "other weight types")
                    res["other_types"] = {}
                    for ot in ("q4_0", "q8_0"):
                        try:
                            res["other_types"][ot] = other_type_decode(pkg, cfg, WTYPES[ot], prompt)
                            res["other_types"][ot + "_free_order"] = other_type_decode(pkg, cfg, WTYPES[ot], prompt, free_order=True)
                        except Exception as e:      # noqa: BLE001
                            res["other_types"][ot] = {"error": str(e)}
                if not args.no_prefill and args.model == "llama3-8b":
                    try:
                        res["prefill"] = prefill_cfg3(pkg, args.model)
                    except Exception as e:
                        res["prefill"] = {"error": str(e)}
        if json_fd is not None:
            os.write(json_fd, (json.dumps(res) + "\n").encode())
        else:
            print(json.dumps(res), flush=True)
    if m is not None:
        m.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
