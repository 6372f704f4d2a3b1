"""The drop-in boundary end to end: the UNMODIFIED reference host (oracle/_ref/ref_chat = chatllm.cpp's model zoo, graph
builder and ggml scheduler, compiled from /root/reference) runs the same synthetic GGMM model twice -- on its own CPU
backend and with every layer offloaded (`-ngl all`) to our libggml-hip.so module -- and the two runs must agree."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
REF = os.path.join(ROOT, "oracle", "_ref")


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
@pytest.mark.parametrize("wname,wt", [("q4_k", 12), ("q8_0", 8), ("q4_0", 2), ("q4_1", 3)])
def test_reference_host_cpu_vs_our_module(gpu, tmp_path, wname, wt):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("tiny", max_len=64)
    mp = str(tmp_path / "m.bin")
    make_ggmm.write_model(mp, cfg, wt, seed=77)
    prompt = [3, 100, 45, 260, 17, 9, 201]
    n_dec = 24

    def run(ngl, teacher=None):
        lp = str(tmp_path / f"l_{ngl}.bin")
        env = dict(os.environ)
        if teacher is not None:
            tf = str(tmp_path / "teacher.txt")
            open(tf, "w").write(" ".join(str(t) for t in teacher))
            env["TEACHER"] = tf
        env["CLLM_HIP_STATS"] = "1"
        r = subprocess.run([os.path.join(REF, "ref_chat"), mp, ngl, "4", str(n_dec), lp] + [str(p) for p in prompt], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return [int(t) for t in r.stdout.split()], np.fromfile(lp, np.float32).reshape(n_dec + 1, cfg["vocab"]), r.stderr

    ids_c, lg_c, _ = run("cpu")
    ids_g, lg_g, err = run("all")                         # FREE-RUNNING: each run feeds its own argmax back
    assert "HIP0" in err or "ggml-hip" in err, err[-500:]      # the module was really loaded (CLLM_HIP_STATS lines)
    assert ids_c == ids_g                                 # greedy token ids identical ...
    assert np.array_equal(lg_c.view(np.uint32), lg_g.view(np.uint32))      # ... and every logit of every step has the same bits


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
@pytest.mark.parametrize("wname,wt,over", [("q4_k", 12, {}), ("q4_0", 2, {}), ("q4_1", 3, {}), ("q4_k", 12, dict(ffn=544))])
def test_module_node_fusion_does_not_change_a_bit(gpu, tmp_path, wname, wt, over):
    """graph_compute fuses RMS_NORM->MUL->MUL_MAT, SILU->MUL->MUL_MAT, MUL_MAT->ADD, SCALE->MASK->SOFT_MAX and the single-token attention
    block into single launches and merges mat-vecs over packed weights; with CLLM_HIP_NO_FUSE=1 every node is its own launch: the logits
    of prompt + 10 decode steps must be identical bytes at every level"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("tiny", max_len=64, **over)
    mp = str(tmp_path / "m.bin")
    make_ggmm.write_model(mp, cfg, wt, seed=78)
    prompt = [3, 100, 45, 260, 17]
    out = {}
    # fused: everything (attention block at level 2 = RoPE + cache writes + attention in one call); attn1: attention without the RoPE /
    # cache-write nodes; attn0: mat-vec and soft_max patterns only; nodes: one call per node
    # nopack: no merged launches over row-repacked weight copies (q|k|v, gate/up)
    # graph: a decode step's launch list is captured the second time it comes and replayed from then on (opt-in)
    # stage: every fused mat-vec / level-1 attention writes to module scratch and is copied into place -- the path taken when ggml-alloc hands a
    # fused launch an output block that overlaps one of its inputs (a freed parent's memory); nograph: every call issued, no launch-list replay
    for mode, extra in (("fused", {}), ("nograph", {"CLLM_HIP_GRAPH": "0"}), ("nopack", {"CLLM_HIP_PACK": "0"}), ("stage", {"CLLM_HIP_FORCE_STAGE": "1", "CLLM_HIP_PACK": "0"}),
                        ("stage1", {"CLLM_HIP_FORCE_STAGE": "1", "CLLM_HIP_FUSE_ATTN": "1"}), ("attn1", {"CLLM_HIP_FUSE_ATTN": "1"}), ("attn0", {"CLLM_HIP_FUSE_ATTN": "0"}),
                        ("nodes", {"CLLM_HIP_NO_FUSE": "1"})):
        lp = str(tmp_path / f"l_{mode}.bin")
        env = dict(os.environ, CLLM_HIP_STATS="1", **extra)
        r = subprocess.run([os.path.join(REF, "ref_chat"), mp, "all", "4", "10", lp] + [str(p) for p in prompt], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        stats = [ln for ln in r.stderr.splitlines() if "graph_compute:" in ln and "calls" in ln]
        out[mode] = (r.stdout.split(), open(lp, "rb").read(), stats)
    n_layer = cfg["n_layer"]
    assert any(f"level 2: {n_layer}," in ln for ln in out["fused"][2]), out["fused"][2][-3:]          # the patterns were really taken
    assert any(f"level 1: {n_layer}," in ln for ln in out["attn1"][2]), out["attn1"][2][-3:]
    assert all("level 1: 0, level 2: 0," in ln for ln in out["attn0"][2])
    import re
    merged = max(int(re.search(r"(\d+) merged", ln).group(1)) for ln in out["fused"][2])
    assert merged == 2 * n_layer if not over else merged >= n_layer, out["fused"][2][-3:]             # q|k|v and gate/up of every layer
    assert all(" 0 merged" in ln for ln in out["nopack"][2])
    assert sum("replayed from the captured graph" in ln for ln in out["fused"][2]) >= 7, out["fused"][2]     # 10 decode steps: issue, capture, then replays (the default)
    assert not any("replayed" in ln for ln in out["nograph"][2])
    for mode in ("fused", "nograph", "nopack", "stage", "stage1", "attn1", "attn0"):
        assert out[mode][0] == out["nodes"][0], mode
        assert out[mode][1] == out["nodes"][1], mode


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
@pytest.mark.parametrize("wname,wt", [("q4_k", 12), ("q8_0", 8)])
def test_reference_host_mixtral_cpu_vs_our_module(gpu, tmp_path, wname, wt):
    """BASELINE cfg5's architecture end to end: a synthetic Mixtral (8 experts, top 2, sliding-window attention class) through the
    unmodified host -- router mat-vec, SOFT_MAX, TOP_K, GET_ROWS, SUM_ROWS, DIV, three MUL_MAT_ID, MUL, ADD of strided views -- on its
    CPU backend and with every layer on our module; every node of the decode graph must run on the device (one graph_compute per token)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("tiny", max_len=64)
    mp = str(tmp_path / "mx.bin")
    make_ggmm.write_mixtral(mp, cfg, wt, seed=91)
    prompt = [5, 9, 42, 300, 7, 99, 250]
    n_dec = 12

    def run(ngl, teacher=None, **extra):
        lp = str(tmp_path / f"l_{ngl}.bin")
        env = dict(os.environ, CLLM_HIP_STATS="1", **extra)
        if teacher is not None:
            tf = str(tmp_path / "teacher.txt")
            open(tf, "w").write(" ".join(str(t) for t in teacher))
            env["TEACHER"] = tf
        env["CLLM_HIP_STATS"] = "1"
        r = subprocess.run([os.path.join(REF, "ref_chat"), mp, ngl, "4", str(n_dec), lp] + [str(p) for p in prompt], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return [int(t) for t in r.stdout.split()], np.fromfile(lp, np.float32).reshape(n_dec + 1, cfg["vocab"]), r.stderr

    ids_c, lg_c, _ = run("cpu")
    ids_g, lg_g, err = run("all", teacher=ids_c)
    _, lg_n, _ = run("all", teacher=ids_c, CLLM_HIP_NO_FUSE="1")
    assert lg_g.tobytes() == lg_n.tobytes()                              # node fusion (incl. the sliding-window class's attention block) changes no bit
    graphs = [ln for ln in err.splitlines() if "graph_compute:" in ln and "calls" in ln]
    assert sum(f"level 2: {cfg['n_layer']}," in ln for ln in graphs) == len(graphs), graphs[:3]        # K row written by CPY instead of SET_ROWS: matched too
    # the reference feeds this architecture one token per graph (batch_input = false, models/mistral.h:101): prompt + decode graphs,
    # and not one more -- no scheduler split, nothing fell back to the CPU backend
    assert len(graphs) == len(prompt) + n_dec, (len(graphs), graphs[:4])
    assert all(f"MoE routers: {cfg['n_layer']}," in ln for ln in graphs), graphs[:3]       # norm + router mat-vec + SOFT_MAX + TOP_K: one launch per block
    # bit-exact token ids at greedy AND bit-identical logits (the run on the module was teacher-forced on the CPU ids only to share it with the
    # no-fusion run: with identical logits its own argmax is the same id)
    assert np.array_equal(lg_c.view(np.uint32), lg_g.view(np.uint32)), [int(np.sum(lg_c[i].view(np.uint32) != lg_g[i].view(np.uint32))) for i in range(n_dec + 1)]
    assert ids_c == ids_g


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
@pytest.mark.parametrize("wname,wt", [("q4_k", 12), ("q4_0", 2)])
def test_reference_host_qwen2_cpu_vs_our_module(gpu, tmp_path, wname, wt):
    """BASELINE cfg4's architecture (Qwen2: q/k/v biases, NEOX RoPE) through the unmodified host: CPU backend vs our module; the bias ADDs
    ride in the mat-vec epilogues, the attention block fuses at level 2, and fusion changes no bit"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("tiny", max_len=64, qkv_bias=1, rope_mode=2, rope_theta=1e6)
    mp = str(tmp_path / "qw.bin")
    make_ggmm.write_model(mp, cfg, wt, seed=93, arch="qwen2")
    prompt = [5, 9, 42, 300, 7, 99, 250]
    n_dec = 12

    def run(ngl, teacher=None, **extra):
        lp = str(tmp_path / f"l_{ngl}.bin")
        env = dict(os.environ, CLLM_HIP_STATS="1", **extra)
        if teacher is not None:
            tf = str(tmp_path / "teacher.txt")
            open(tf, "w").write(" ".join(str(t) for t in teacher))
            env["TEACHER"] = tf
        env["CLLM_HIP_STATS"] = "1"
        r = subprocess.run([os.path.join(REF, "ref_chat"), mp, ngl, "4", str(n_dec), lp] + [str(p) for p in prompt], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return [int(t) for t in r.stdout.split()], np.fromfile(lp, np.float32).reshape(n_dec + 1, cfg["vocab"]), r.stderr

    ids_c, lg_c, _ = run("cpu")
    ids_g, lg_g, err = run("all", teacher=ids_c)
    _, lg_n, _ = run("all", teacher=ids_c, CLLM_HIP_NO_FUSE="1")
    assert lg_g.tobytes() == lg_n.tobytes()
    graphs = [ln for ln in err.splitlines() if "graph_compute:" in ln and "calls" in ln]
    assert len(graphs) == n_dec + 1 and sum(f"level 2: {cfg['n_layer']}," in ln for ln in graphs) == n_dec, graphs[:3]
    assert sum(f" {2 * cfg['n_layer']} merged" in ln for ln in graphs) == n_dec, graphs[-2:]       # q|k|v (with the packed biases) and gate/up of every layer
    # bit-exact token ids at greedy AND bit-identical logits (the run on the module was teacher-forced on the CPU ids only to share it with the
    # no-fusion run: with identical logits its own argmax is the same id)
    assert np.array_equal(lg_c.view(np.uint32), lg_g.view(np.uint32)), [int(np.sum(lg_c[i].view(np.uint32) != lg_g[i].view(np.uint32))) for i in range(n_dec + 1)]
    assert ids_c == ids_g


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
def test_baseline_cfg1_gpt2_small_sized_q8_0_greedy_64(gpu, tmp_path):
    """BASELINE cfg1 (D1: a GPT-2-small-sized Llama-architecture stand-in, L=12, H=768, 12 heads, F=3072, V=50304, Q8_0): the
    reference host greedy-decodes 64 tokens on its CPU backend; the same host on our module, FREE-RUNNING, produces the same ids and
    bit-identical logits at every step"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("gpt2s-llama", max_len=128)
    mp = str(tmp_path / "g.bin")
    make_ggmm.write_model(mp, cfg, 8, seed=5, fast=True)
    prompt = [3, 100, 45, 260, 17, 9, 201, 4000, 31000]
    n_dec = 64

    def run(ngl, teacher=None):
        lp = str(tmp_path / f"l_{ngl}.bin")
        env = dict(os.environ)
        if teacher is not None:
            tf = str(tmp_path / "teacher.txt")
            open(tf, "w").write(" ".join(str(t) for t in teacher))
            env["TEACHER"] = tf
        r = subprocess.run([os.path.join(REF, "ref_chat"), mp, ngl, "8", str(n_dec), lp] + [str(p) for p in prompt], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return [int(t) for t in r.stdout.split()], np.fromfile(lp, np.float32).reshape(n_dec + 1, cfg["vocab"])

    ids_c, lg_c = run("cpu")
    ids_g, lg_g = run("all")
    # bit-exact token ids at greedy AND bit-identical logits (the run on the module was teacher-forced on the CPU ids only to share it with the
    # no-fusion run: with identical logits its own argmax is the same id)
    assert np.array_equal(lg_c.view(np.uint32), lg_g.view(np.uint32)), [int(np.sum(lg_c[i].view(np.uint32) != lg_g[i].view(np.uint32))) for i in range(n_dec + 1)]
    assert ids_c == ids_g


# ---- BASELINE configs at REAL shapes: the unmodified reference host, CPU backend vs every layer on our module, FREE-RUNNING greedy --------------
def _real_shape_case(gpu, tmp_path, arch, cfgname, wt, prompt, n_dec, over, threads=32):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config(cfgname, **over)
    mp = str(tmp_path / "m.bin")
    if arch == "mixtral":
        make_ggmm.write_mixtral(mp, cfg, wt, seed=5, fast=True)
    else:
        make_ggmm.write_model(mp, cfg, wt, seed=5, fast=True, arch=arch)

    def run(ngl):
        lp = str(tmp_path / f"l_{ngl}.bin")
        r = subprocess.run([os.path.join(REF, "ref_chat"), mp, ngl, str(threads), str(n_dec), lp] + [str(p) for p in prompt], capture_output=True, text=True,
                           env=dict(os.environ, CLLM_HIP_STATS="1"), timeout=1800)
        assert r.returncode == 0, r.stderr[-2000:]
        return [int(t) for t in r.stdout.split()], np.fromfile(lp, np.float32).reshape(n_dec + 1, cfg["vocab"]), r.stderr

    ids_c, lg_c, _ = run("cpu")
    ids_g, lg_g, err = run("all")
    os.remove(mp)
    assert "graph_compute:" in err                                    # the module really computed the graphs
    mism = sum(int(a != b) for a, b in zip(ids_c, ids_g))
    within = float(np.mean(np.max(np.abs(lg_c - lg_g), axis=1) <= 1e-3))
    words = int(np.sum(lg_c.view(np.uint32) != lg_g.view(np.uint32)))
    print(f"{arch}/{cfgname}: greedy id mismatches {mism}/{len(ids_c)}, steps with max|dlogit| <= 1e-3: {within:.3f}, differing logit words {words}")
    assert mism == 0 and within == 1.0          # north star: bit-exact token ids at greedy, logits within 1e-3 -- at every step
    assert words == 0                           # and in fact bit-identical logits (AVX2-order accumulation, glibc-exact libm)


_BIG = pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))) or os.environ.get("CLLM_SKIP_BIG"),
                          reason="oracle/_ref (reference host + module) not built, or CLLM_SKIP_BIG set")


@_BIG
def test_baseline_cfg2_llama3_8b_q4_k_16_plus_64_free_running(gpu, tmp_path):
    """BASELINE cfg2: Llama-3-8B shapes, Q4_K, 16-token prompt, 64 greedy tokens (4.7 GB synthetic GGMM file)"""
    _real_shape_case(gpu, tmp_path, "llama3", "llama3-8b", 12, [(7 * i + 3) % 128000 for i in range(16)], 64, dict(max_len=1024))


@_BIG
@pytest.mark.parametrize("wname,wt", [("q4_0", 2), ("q4_k", 12)])
def test_llama3_8b_512_token_prompt_free_running(gpu, tmp_path, wname, wt):
    """a prompt far beyond the 32-column limit of round 2's exact kernels at Llama-3-8B shapes: 512 tokens as one graph, then 8 greedy tokens, free-running;
    0 id mismatches, every logit word identical (mmx.hip + mmf_exact.hip behind the module's prefill patterns)"""
    _real_shape_case(gpu, tmp_path, "llama3", "llama3-8b", wt, [(7 * i + 11) % 32000 for i in range(512)], 8, dict(max_len=640), threads=64)


@_BIG
@pytest.mark.skipif(bool(os.environ.get("CLLM_SKIP_CFG3")), reason="CLLM_SKIP_CFG3 set (the reference's CPU run of the 4096-token prompt takes ~100 s on the box's host)")
def test_baseline_cfg3_llama3_8b_q4_0_4096_token_prompt_free_running(gpu, tmp_path):
    """BASELINE cfg3: Llama-3-8B shapes, Q4_0, ONE 4096-token prompt (a single graph), then 4 greedy tokens, free-running, CPU host vs every layer on the module"""
    _real_shape_case(gpu, tmp_path, "llama3", "llama3-8b", 2, [(7 * i + 11) % 32000 for i in range(4096)], 4, dict(max_len=4608), threads=64)


@_BIG
def test_baseline_cfg5_mixtral_shapes_q4_k_free_running(gpu, tmp_path):
    """BASELINE cfg5's block at real shapes (Mixtral-8x7B: H 4096, F 14336, 8 experts, top 2, Q4_K), 4 of the 32 layers to bound the file (3.4 GB)"""
    _real_shape_case(gpu, tmp_path, "mixtral", "mixtral-8x7b", 12, [(11 * i + 5) % 32000 for i in range(16)], 48, dict(max_len=512, n_layer=4))


@_BIG
def test_baseline_cfg4_qwen2_72b_shapes_q4_k_free_running(gpu, tmp_path):
    """BASELINE cfg4's block at real shapes (Qwen2-72B: H 8192, 64 heads / 8 kv, F 29568 with the Q8_0 down_proj, q/k/v biases, NEOX RoPE), 2 of the
    80 layers to bound the file (2.6 GB incl. the 152064-row embedding and lm_head)"""
    _real_shape_case(gpu, tmp_path, "qwen2", "qwen2-72b", 12, [(13 * i + 7) % 150000 for i in range(16)], 32, dict(max_len=512, n_layer=2))


_REF_BUILT = pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                                reason="oracle/_ref (reference host + module) not built")


@_REF_BUILT
def test_full_depth_80_layers_qwen2_arch_reduced_width_free_running(gpu, tmp_path):
    """BASELINE cfg4's DEPTH in every run of the suite: all 80 layers of the Qwen2 architecture (q/k/v biases, NEOX RoPE, GQA 8 : 2) at reduced width -- hidden 1024,
    ffn 3 * 256 + 32 = 800, so the Q4_K file's down_proj still falls back to Q8_0 as Qwen2-72B's 29568 does (convert.py:811-829) -- Q4_K, 230 MB.  The reference host on
    its CPU backend vs every layer on the module, free-running greedy over a 16-token prompt + 24 tokens: equal ids, zero differing logit words.  HeterogeneousModel::forward
    walks all layers (src/models.cpp:1399-1424): an error that compounds with depth (a residual stream kept in the wrong buffer, a KV cache of the wrong layer, the
    launch list of one layer replayed for another) shows here; the real-width cases above run 2 of the 80."""
    _real_shape_case(gpu, tmp_path, "qwen2", "qwen2-72b", 12, [(13 * i + 7) % 2000 for i in range(16)], 24,
                     dict(max_len=256, hidden=1024, n_head=8, n_kv_head=2, ffn=800, vocab=2048), threads=8)


@_REF_BUILT
def test_full_depth_32_layers_mixtral_arch_reduced_width_free_running(gpu, tmp_path):
    """BASELINE cfg5's DEPTH in every run of the suite: all 32 layers of the Mixtral architecture (8 experts, top 2, one token per graph through the prompt) at
    hidden 1024 / ffn 768, Q4_K, 390 MB: CPU host vs every layer on the module, free-running, equal ids and zero differing logit words over 16 + 24 steps."""
    _real_shape_case(gpu, tmp_path, "mixtral", "mixtral-8x7b", 12, [(11 * i + 5) % 2000 for i in range(16)], 24,
                     dict(max_len=256, hidden=1024, n_head=8, n_kv_head=2, ffn=768, vocab=2048), threads=8)


_FULL = pytest.mark.skipif(not os.environ.get("CLLM_FULL_DEPTH"), reason="CLLM_FULL_DEPTH=1 runs BASELINE cfg4 / cfg5 at FULL depth (26 / 50 GB GGMM files in /tmp, minutes of "
                                                                    "CPU time for the reference's own run); their last record: profiles/r05_full_depth_parity_cfg4_cfg5.txt")


@_BIG
@_FULL
@pytest.mark.parametrize("arch,cname", [("mixtral", "mixtral-8x7b"), ("qwen2", "qwen2-72b")])
def test_baseline_cfg4_cfg5_full_depth_free_running(gpu, arch, cname):
    """BASELINE cfg5 (Mixtral-8x7B shapes: 32 layers, 8 experts, top 2) and cfg4 (Qwen2-72B shapes: 80 layers, Q8_0 down_proj, biases, NEOX RoPE) at FULL depth,
    Q4_K: the reference host on its CPU backend vs every layer on the module, free-running greedy over the 16-token prompt + 16 tokens -- equal ids, zero differing
    logit words (src/models.cpp:1399-1424 walks all layers; the block-level cases above cover 4 and 2 of them)"""
    sys.path.insert(0, ROOT)
    import bench
    mp = f"/tmp/{cname}-q4_k.bin"
    if not os.path.exists(mp):
        rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_ggmm.py"), "--arch", arch, "--config", cname, "--wtype", "q4_k", "--max-len", "512", "--fast", "--out", mp],
                            capture_output=True, text=True)
        assert rc.returncode == 0, rc.stderr[-500:]
    p = bench.parity_full_depth(mp, gpu.synth.config(cname, max_len=512)["vocab"], n_decode=16)
    print(f"{cname} full depth: {p}")
    assert p["ids_equal"] and p["logit_words_differing"] == 0, p


def _host_run(tmp_path, mp, ngl, n_dec, prompt, vocab, teacher=None, threads=4, **extra):
    lp = str(tmp_path / f"l_{ngl}_{len(extra)}.bin")
    env = dict(os.environ, CLLM_HIP_STATS="1", **extra)
    if teacher is not None:
        tf = str(tmp_path / "teacher.txt")
        open(tf, "w").write(" ".join(str(t) for t in teacher))
        env["TEACHER"] = tf
    r = subprocess.run([os.path.join(REF, "ref_chat"), mp, ngl, str(threads), str(n_dec), lp] + [str(p) for p in prompt], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [int(t) for t in r.stdout.split()], np.fromfile(lp, np.float32).reshape(n_dec + 1, vocab), r.stderr


def _tolerance_tier(lg_c, lg_g, ids_c, tol):
    """teacher-forced comparison of a tolerance-tier path: every step's logits within tol * std(logits), and the argmax agrees wherever
    the CPU's top-1 margin exceeds twice the deviation actually observed at that step"""
    sigma = float(np.std(lg_c))
    dev = np.max(np.abs(lg_c - lg_g), axis=1)
    assert float(np.max(dev)) < tol * sigma, (float(np.max(dev)), sigma)
    top2 = np.sort(lg_c, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2 * dev
    assert np.all(np.argmax(lg_g, axis=1)[clear] == np.argmax(lg_c, axis=1)[clear])
    return float(np.max(dev)) / sigma, float(np.mean(clear))


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
@pytest.mark.parametrize("cache", ["f16", "q8_0"])
def test_reference_host_flash_attention_on_our_module(gpu, tmp_path, cache):
    """`-fa 1` (src/layers.cpp:2634-2656) with --cache_dtype f16 | q8_0: FLASH_ATTN_EXT, the F16 run-time mask and the SET_ROWS into
    quantized cache rows all stay on the device; tolerance tier (the CPU op changes summation order with shape and thread count)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("small", max_len=128)
    mp = str(tmp_path / "m.bin")
    make_ggmm.write_model(mp, cfg, 12, seed=79)
    prompt = [(11 * i + 5) % cfg["vocab"] for i in range(24)]        # 24 query rows (the mat-muls stay on the exact kernels: what differs is the attention), then 20 single-token steps
    n_dec = 20
    ids_c, lg_c, _ = _host_run(tmp_path, mp, "cpu", n_dec, prompt, cfg["vocab"], REF_CHAT_FA="1", REF_CHAT_CACHE=cache)
    ids_g, lg_g, err = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"], teacher=ids_c, REF_CHAT_FA="1", REF_CHAT_CACHE=cache)
    stats = [ln for ln in err.splitlines() if "graph_compute:" in ln and "calls" in ln]
    assert stats and all(f"flash_attn_ext: {cfg['n_layer']}" in ln for ln in stats), stats[-2:]      # every layer's node ran on the module
    # the yardstick: the reference against itself -- the same model, same tokens, its eager attention instead of its flash attention (both on
    # its CPU backend).  Quantized activations amplify a 1e-3 difference in an attention output into O(0.1 sigma) logit differences.
    _, lg_e, _ = _host_run(tmp_path, mp, "cpu", n_dec, prompt, cfg["vocab"], teacher=ids_c, REF_CHAT_CACHE=cache)
    dev_ref, _ = _tolerance_tier(lg_c, lg_e, ids_c, 0.5)
    dev, clear = _tolerance_tier(lg_c, lg_g, ids_c, 0.3)
    print(f"flash attention, cache {cache}: module vs CPU max|dlogit| = {dev:.2e} sigma (steps with a clear top-1: {clear:.2f}); CPU flash vs CPU eager: {dev_ref:.2e} sigma")
    assert dev < 2.0 * dev_ref + 0.02


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
@pytest.mark.parametrize("wname,wt,nprompt", [("q4_0", 2, 70), ("q4_k", 12, 200), ("q8_0", 8, 45), ("q4_1", 3, 33)])
def test_reference_host_long_prompt_is_bit_identical_free_running(gpu, tmp_path, wname, wt, nprompt):
    """the default (eager) attention with a prompt of more than 32 tokens, default prefill mode: the quantized mat-muls run on mmx.hip (with the module's
    prologue / epilogue fusions), MUL_MAT + SCALE + DIAG_MASK_INF + SOFT_MAX + MUL_MAT of every layer as one cllm_op_attn_prefill call on the exact-order
    kernels -- FREE-RUNNING, every logit of the prompt and of the decode steps after it has the bits of the reference's CPU run"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("small", max_len=256)
    mp = str(tmp_path / "m.bin")
    make_ggmm.write_model(mp, cfg, wt, seed=80)
    prompt = [(7 * i + 3) % cfg["vocab"] for i in range(nprompt)]
    ids_c, lg_c, _ = _host_run(tmp_path, mp, "cpu", 8, prompt, cfg["vocab"])
    ids_g, lg_g, err = _host_run(tmp_path, mp, "all", 8, prompt, cfg["vocab"])
    stats = [ln for ln in err.splitlines() if "graph_compute:" in ln and "calls" in ln]
    assert f"flash prefill: {cfg['n_layer']}" in stats[0], stats[0]           # the attention pattern was taken (one call per layer)
    assert f"prefill mat-muls with fused prologue / epilogue: {7 * cfg['n_layer']})" in stats[0], stats[0]
    assert ids_c == ids_g
    assert np.array_equal(lg_c.view(np.uint32), lg_g.view(np.uint32)), [int(np.sum(lg_c[i].view(np.uint32) != lg_g[i].view(np.uint32))) for i in range(9)]
    # and the same bits without the module's prefill patterns (one call per node)
    ids_p, lg_p, _ = _host_run(tmp_path, mp, "all", 8, prompt, cfg["vocab"], CLLM_HIP_NO_PREFILL_FUSE="1", CLLM_HIP_FUSE_ATTN="0")
    assert lg_g.tobytes() == lg_p.tobytes()


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
@pytest.mark.parametrize("arch,wname,wt,nprompt", [("llama", "q4_k", 12, 10), ("llama", "q4_0", 2, 12), ("llama", "q4_k", 12, 20), ("llama", "q8_0", 8, 32), ("llama", "q4_1", 3, 17),
                                                   ("qwen2", "q4_k", 12, 70), ("qwen2", "q4_0", 2, 24), ("qwen2", "q8_0", 8, 140)])
def test_reference_host_prompts_of_every_length_class_are_bit_identical_free_running(gpu, tmp_path, arch, wname, wt, nprompt):
    """prompts between the exact GEMM's take-over (5 / 10 columns) and 32 tokens -- mat-muls on mmx.hip with the module's prompt patterns, the attention still node by
    node -- and longer ones on the Qwen2 architecture (q / k / v biases, NEOX RoPE): FREE-RUNNING, every logit of the prompt and of the decode steps after it has the
    bits of the reference's CPU run"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    over = dict(qkv_bias=1, rope_mode=2, rope_theta=1e6) if arch == "qwen2" else {}
    cfg = gpu.synth.config("small", max_len=256, **over)
    mp = str(tmp_path / "m.bin")
    make_ggmm.write_model(mp, cfg, wt, seed=81, **({"arch": "qwen2"} if arch == "qwen2" else {}))
    prompt = [(11 * i + 5) % cfg["vocab"] for i in range(nprompt)]
    ids_c, lg_c, _ = _host_run(tmp_path, mp, "cpu", 8, prompt, cfg["vocab"])
    ids_g, lg_g, _ = _host_run(tmp_path, mp, "all", 8, prompt, cfg["vocab"])
    assert ids_c == ids_g
    assert np.array_equal(lg_c.view(np.uint32), lg_g.view(np.uint32)), [int(np.sum(lg_c[i].view(np.uint32) != lg_g[i].view(np.uint32))) for i in range(9)]


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
def test_reference_host_long_prompt_fast_mode_uses_the_flash_prefill(gpu, tmp_path):
    """CLLM_PREFILL=fast: MUL_MAT + SCALE + DIAG_MASK_INF + SOFT_MAX + MUL_MAT of every layer run as one flash kernel and the mat-muls on the int8 MFMA GEMM
    (tolerance tier); the decode steps after it are the exact kernels again"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("small", max_len=128)
    mp = str(tmp_path / "m.bin")
    make_ggmm.write_model(mp, cfg, 2, seed=80)
    prompt = [(7 * i + 3) % cfg["vocab"] for i in range(70)]
    ids_c, lg_c, _ = _host_run(tmp_path, mp, "cpu", 8, prompt, cfg["vocab"])
    ids_g, lg_g, err = _host_run(tmp_path, mp, "all", 8, prompt, cfg["vocab"], teacher=ids_c, CLLM_PREFILL="fast")
    stats = [ln for ln in err.splitlines() if "graph_compute:" in ln and "calls" in ln]
    assert f"flash prefill: {cfg['n_layer']}" in stats[0], stats[0]
    assert all("flash prefill: 0" in ln for ln in stats[1:])
    dev, clear = _tolerance_tier(lg_c, lg_g, ids_c, 0.25)
    ids_n, lg_n, _ = _host_run(tmp_path, mp, "all", 8, prompt, cfg["vocab"], teacher=ids_c, CLLM_PREFILL="fast", CLLM_FLASH_PREFILL="0")
    dev_n, _ = _tolerance_tier(lg_c, lg_n, ids_c, 0.25)
    print(f"70-token prompt, fast mode: flash prefill max|dlogit| = {dev:.2e} sigma, node sequence (MFMA mat-muls) {dev_n:.2e} sigma")
    assert dev < 2 * dev_n + 1e-3                                    # the fused form is no worse than the node sequence it replaces
    # the prompt graph's quantized mat-muls carry their neighbours: norm / SiLU * up in the quantizer, q / k / v and gate / up share one quantization, residual
    # adds in the epilogue (7 per layer) -- and that changes no bit against the same mat-muls issued node by node
    assert f"prefill mat-muls with fused prologue / epilogue: {7 * cfg['n_layer']})" in stats[0], stats[0]
    ids_p, lg_p, _ = _host_run(tmp_path, mp, "all", 8, prompt, cfg["vocab"], teacher=ids_c, CLLM_PREFILL="fast", CLLM_HIP_NO_PREFILL_FUSE="1")
    assert lg_g.tobytes() == lg_p.tobytes()


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_backend_async")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference ggml + module) not built")
def test_module_events_and_async_copy_between_backends(gpu):
    """ggml_backend_i.cpy_tensor_async / event_record / event_wait and the device's event_new / event_synchronize (ggml-backend-impl.h:87-127),
    driven through the reference's public ggml-backend API the way its scheduler does at a layer split: two backends (streams), a -> b -> a"""
    r = subprocess.run([os.path.join(REF, "ref_backend_async"), os.path.join(REF, "libggml-hip.so"), str(3 << 20)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK"), (r.stdout, r.stderr[-1500:])


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
@pytest.mark.parametrize("name,wt,mix", [("q6_k", 14, None), ("q5_k", 13, None), ("q4_k_m", 12, {"wv": 14, "wdown": 14, "lm_head": 14, "tok_embd": 14, "wo": 13})])
def test_reference_host_other_k_quants_bit_identical(gpu, tmp_path, name, wt, mix):
    """the k-quants third-party GGMM files carry besides Q4_K -- Q5_K, Q6_K, and a Q4_K_M-style mix (Q6_K embedding / output / v / down, Q5_K o):
    every node stays on the module and the free-running generation is bit-identical to the reference's CPU run"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("tiny", max_len=64, mix=mix)
    mp = str(tmp_path / "m.bin")
    make_ggmm.write_model(mp, cfg, wt, seed=81)
    prompt = [3, 100, 45, 260, 17, 9, 201]
    ids_c, lg_c, _ = _host_run(tmp_path, mp, "cpu", 16, prompt, cfg["vocab"])
    ids_g, lg_g, err = _host_run(tmp_path, mp, "all", 16, prompt, cfg["vocab"])
    graphs = [ln for ln in err.splitlines() if "graph_compute:" in ln and "calls" in ln]
    assert len(graphs) == 17, len(graphs)                  # one graph per step: nothing fell back to the CPU backend
    assert ids_c == ids_g
    assert np.array_equal(lg_c.view(np.uint32), lg_g.view(np.uint32))


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
@pytest.mark.parametrize("name,wt,mix,nprompt", [
    ("legacy mix", 2, {"wq": 6, "wk": 7, "wv": 20, "wo": 11, "wgate": 10, "wup": 6, "wdown": 7, "lm_head": 20, "tok_embd": 6}, 7),       # Q5_0 / Q5_1 / IQ4_NL / Q3_K / Q2_K
    ("legacy mix, long prompt", 2, {"wq": 6, "wk": 7, "wv": 20, "wo": 11, "wgate": 10, "wup": 6, "wdown": 7, "lm_head": 20, "tok_embd": 6}, 50),
    ("q5_0", 6, None, 9), ("q5_1", 7, None, 9), ("iq4_nl", 20, None, 40), ("q2_k", 10, None, 9), ("q3_k", 11, None, 40), ("mxfp4", 39, None, 9), ("iq4_xs", 23, None, 40),
    ("tq2_0", 35, None, 9), ("tq1_0", 34, None, 40), ("ternary mix", 35, {"wq": 34, "wo": 34, "wdown": 34, "lm_head": 35, "tok_embd": 34}, 9),
    ("iq2_xxs", 16, None, 9), ("iq2_xs", 17, None, 40), ("iq2_s", 22, None, 9), ("iq3_xxs", 18, None, 9), ("iq3_s", 21, None, 40), ("iq1_s", 19, None, 9), ("iq1_m", 29, None, 40),
    ("codebook mix", 2, {"wq": 16, "wk": 17, "wv": 22, "wo": 21, "wgate": 17, "wup": 22, "wdown": 18, "lm_head": 22, "tok_embd": 17}, 9)])      # (a codebook BASE type with differing per-tensor
    # types makes the reference host re-quantize into it, which needs ggml_quantize_init: it aborts on its own CPU backend too)
def test_reference_host_other_formats_bit_identical(gpu, tmp_path, name, wt, mix, nprompt):
    """model files in the formats older and third-party converters write -- Q5_0, Q5_1, IQ4_NL, IQ4_XS, Q2_K, Q3_K, MXFP4, TQ1_0, TQ2_0, IQ2_XXS, IQ2_XS, IQ2_S, IQ3_XXS, IQ3_S, IQ1_S, IQ1_M, pure and mixed per tensor -- through the unmodified
    host: every node stays on the module (no CPU fallback) and the free-running generation has the bits of the reference's CPU run, for one-token graphs, short
    prompts (one-column order) and long ones (IQ4_NL's other order under tinyBLAS)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("tiny", max_len=128, mix=mix)
    mp = str(tmp_path / "m.bin")
    make_ggmm.write_model(mp, cfg, wt, seed=85)
    prompt = [(13 * i + 3) % cfg["vocab"] for i in range(nprompt)]
    ids_c, lg_c, _ = _host_run(tmp_path, mp, "cpu", 10, prompt, cfg["vocab"])
    ids_g, lg_g, err = _host_run(tmp_path, mp, "all", 10, prompt, cfg["vocab"])
    graphs = [ln for ln in err.splitlines() if "graph_compute:" in ln and "calls" in ln]
    assert len(graphs) == 11, len(graphs)                  # one graph per step: nothing fell back to the CPU backend
    assert ids_c == ids_g
    assert np.array_equal(lg_c.view(np.uint32), lg_g.view(np.uint32))


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
def test_reference_host_mixtral_with_q6_k_and_q5_k_experts_bit_identical(gpu, tmp_path):
    """a Q4_K_M-style Mixtral file: expert down projections in Q6_K, expert up projections in Q5_K, the rest Q4_K -- MUL_MAT_ID over coverage-type experts runs on the
    module (one grid slice per (token, slot)), nothing falls back to the CPU, the bits are the CPU run's"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("tiny", max_len=64, rope_theta=1e6, mix={".w2.": 14, ".w3.": 13})
    mp = str(tmp_path / "mx.bin")
    make_ggmm.write_mixtral(mp, cfg, 12, seed=92)
    prompt = [5, 9, 42, 300, 7]
    ids_c, lg_c, _ = _host_run(tmp_path, mp, "cpu", 10, prompt, cfg["vocab"])
    ids_g, lg_g, err = _host_run(tmp_path, mp, "all", 10, prompt, cfg["vocab"])
    assert ids_c == ids_g
    assert np.array_equal(lg_c.view(np.uint32), lg_g.view(np.uint32))
    graphs = [ln for ln in err.splitlines() if "graph_compute:" in ln and "calls" in ln]
    assert len(graphs) == len(prompt) + 10, len(graphs)    # (this architecture feeds the prompt one token per graph) one graph per step: no split to the CPU backend


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
def test_module_decode_ahead_hits_and_misses_stay_bit_identical(gpu, tmp_path):
    """decode-ahead (ggml-hip.cpp ahead_launch): after the host has read a step's logits the module starts the next greedy step itself; when the
    host's graph arrives it is compared with the prediction.  Free-running greedy: every step is a hit.  A teacher that feeds OTHER tokens than the
    argmax: every step is a miss (the step runs again the normal way, with the host's scalars written after the wrong one finished).  Either way every
    logit equals the CPU run's bits, and the module counts what it did."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    import re
    cfg = gpu.synth.config("tiny", max_len=128)
    mp = str(tmp_path / "m.bin")
    make_ggmm.write_model(mp, cfg, 12, seed=83)
    prompt = [3, 100, 45, 260, 17]
    n_dec = 100
    ids_c, lg_c, _ = _host_run(tmp_path, mp, "cpu", n_dec, prompt, cfg["vocab"])
    ids_g, lg_g, err = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"])
    assert ids_c == ids_g and np.array_equal(lg_c.view(np.uint32), lg_g.view(np.uint32))
    m = re.findall(r"steps started ahead of the host: (\d+), of which the host then asked for: (\d+)", err)
    assert m and int(m[-1][0]) >= 50 and int(m[-1][1]) >= int(m[-1][0]) - 1, m                        # greedy: (almost) every step was a hit
    ids_off, lg_off, _ = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"], CLLM_HIP_AHEAD="0")
    assert np.array_equal(lg_off.view(np.uint32), lg_c.view(np.uint32))
    teacher = [(37 * i + 11) % cfg["vocab"] for i in range(n_dec + 1)]                               # not the argmax: every prediction is wrong
    _, lt_c, _ = _host_run(tmp_path, mp, "cpu", n_dec, prompt, cfg["vocab"], teacher=teacher)
    _, lt_g, err = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"], teacher=teacher)
    assert np.array_equal(lt_c.view(np.uint32), lt_g.view(np.uint32))
    m = re.findall(r"steps started ahead of the host: (\d+), of which the host then asked for: (\d+)", err)
    assert m and int(m[-1][1]) <= 2 and 1 <= int(m[-1][0]) <= 12, m                                   # two misses in a row switch it off for 64 graphs


# ---- the boundary between `cpu` and `all`: partial offload and the reference's layer split (src/backend.cpp:677-778), through the real scheduler ----------------
_SPLITS = [
    # (CLLM_HIP_VIRTUAL_DEVICES, -ngl spec)
    (0, "1"),                              # layer 0 on the module; layers 1..3, embedding and lm_head on the CPU: the residual crosses twice per graph
    (0, "2,prolog"),                       # embedding + the first two layers
    (0, "3,epilog"),                       # three layers + final norm / lm_head
    (0, "prolog,epilog"),                  # only the embedding and the head on the module (every decoder layer on the CPU)
    (2, "0:2;1:2"),                        # the layer split over TWO of the module's devices (both on this GPU), embedding / head on the CPU
    (2, "0:2,prolog;1:2,epilog"),          # ... everything on the two devices: the residual crosses once, device to device (cpy_tensor_async + event)
    (2, "0:1;1:2"),                        # two devices AND a CPU remainder
    (3, "0:1,prolog;1:2;2:1,epilog"),      # three devices
]


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ref_chat")) and os.path.exists(os.path.join(REF, "libggml-hip.so"))),
                    reason="oracle/_ref (reference host + module) not built")
@pytest.mark.parametrize("wname,wt", [("q4_k", 12), ("q4_0", 2)])
def test_reference_host_partial_offload_and_layer_split_are_bit_identical(gpu, tmp_path, wname, wt):
    """SURVEY 7.2's minimum slice: MIXED CPU / GPU runs.  The unmodified host places some layers on our module and the rest on its CPU backend (`-ngl 1`,
    `-ngl 2,prolog` ...), or splits them over several of the module's devices (`-ngl "0:2;1:2"`: CLLM_HIP_VIRTUAL_DEVICES registers n ggml devices on the one
    GPU, each with its own buffer type, backend and stream, so ggml's scheduler cuts the graph, copies the residual across and orders the streams exactly
    as on n GPUs).  Every placement must reproduce the CPU run: free-running greedy ids and every logit word, for a prompt (one multi-token graph) and the
    single-token graphs after it."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("tiny", max_len=96, n_layer=4)
    mp = str(tmp_path / "m4.bin")
    make_ggmm.write_model(mp, cfg, wt, seed=404)
    prompt = [(37 * i + 11) % cfg["vocab"] for i in range(19)]
    n_dec = 14
    ids_c, lg_c, _ = _host_run(tmp_path, mp, "cpu", n_dec, prompt, cfg["vocab"])
    ids_a, lg_a, _ = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"])
    assert ids_c == ids_a and np.array_equal(lg_c.view(np.uint32), lg_a.view(np.uint32))
    for k, (nvirt, spec) in enumerate(_SPLITS):
        extra = {"CLLM_HIP_VIRTUAL_DEVICES": str(nvirt)} if nvirt else {}
        ids_s, lg_s, err = _host_run(tmp_path, mp, spec, n_dec, prompt, cfg["vocab"], SPLIT_CASE=str(k), **extra)
        assert "ggml-hip" in err or "HIP0" in err, (spec, err[-300:])          # the module took part
        if nvirt:
            assert f"HIP{nvirt - 1}" in err, (spec, err[-600:])                # ... with its last device in use
        assert ids_s == ids_c, (spec, ids_s, ids_c)
        assert np.array_equal(lg_s.view(np.uint32), lg_c.view(np.uint32)), (spec, int(np.sum(lg_s.view(np.uint32) != lg_c.view(np.uint32))))


@_REF_BUILT
@pytest.mark.parametrize("arch,wt,over", [("llama3", 12, dict(hidden=2048, n_head=16, n_kv_head=8, head_dim=128, ffn=2816, vocab=2048, n_layer=3)),
                                          ("qwen2", 12, dict(hidden=2048, n_head=16, n_kv_head=8, head_dim=128, ffn=1824, vocab=2048, n_layer=2, qkv_bias=1, rope_mode=2, rope_theta=1e6)),
                                          ("llama3", 2, dict(hidden=512, n_head=8, n_kv_head=8, head_dim=64, ffn=1056, vocab=1024, n_layer=2))])
def test_tensor_parallel_behind_the_ggml_boundary(gpu, tmp_path, arch, wt, over):
    """CLLM_HIP_TP=N: the UNMODIFIED host sees ONE ggml device; the module shards every decode step over N ranks behind it (host/ggml-hip.cpp tp_graph_compute -- the reference's
    own slot is SplitMethod::Row, "TODO: WIP", src/backend.h:322-327).  N = 2, 4, 8 virtual ranks on the one GPU of the box: weight shards cut on the device from the tensors the
    host uploaded (q|k|v / gate|up by rows, o / down by whole quant blocks: ffn 2816 = 11 Q4_K blocks -> 8 ranks get 2,2,2,1,1,1,1,1; Qwen2's Q8_0 down_proj: 57 blocks of 32;
    Q4_0: 33 blocks), one KV shard per rank refreshed from the host's cache after the un-sharded prompt and written back every step, the all-reduce fused into the mat-vecs.
    Tolerance tier (the fp32 sums of o / down become N partial chains): teacher-forced logits within 0.25 sigma of the single-device run, argmax equal wherever the margin is
    clear; then a second prompt (un-sharded, attends over the cache rows the sharded steps wrote back) and more sharded steps in the same process."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("small", max_len=128, **over)
    mp = str(tmp_path / "m.bin")
    make_ggmm.write_model(mp, cfg, wt, seed=31, fast=True, arch=arch)
    prompt = [(17 * i + 3) % cfg["vocab"] for i in range(9)]
    n_dec = 14
    turn2 = dict(REF_CHAT_CHUNK_AT="8", REF_CHAT_CHUNK=" ".join(str((29 * i + 11) % cfg["vocab"]) for i in range(5)))      # step 8: its token + 5 more ids as one graph
    ids_1, lg_1, err_1 = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"], threads=4, **turn2)
    assert "tensor parallel" not in err_1
    for n in (2, 4, 8):
        ids_n, lg_n, err = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"], teacher=ids_1, threads=4, CLLM_HIP_TP=str(n), CLLM_HIP_TP_DEBUG="1", **turn2)
        assert f"tensor parallel: {n} ranks behind one ggml device" in err, err[-1500:]
        steps = [ln for ln in err.splitlines() if "-> tensor parallel over" in ln]
        assert len(steps) == n_dec - 1, (len(steps), err[-1500:])                 # every single-token step ran sharded; the prompt and the second turn's chunk did not
        assert sum("replayed from the captured graphs" in ln or "started ahead of the host" in ln for ln in steps) >= 4, steps[-3:]      # the sharded step's launch list is captured (2nd identical step) and replayed -- by the host's request or ahead of it --, before and after the chunk
        assert all("lm_head rows sharded" in ln for ln in steps)
        assert "timed out" not in err
        assert np.array_equal(lg_1[0].view(np.uint32), lg_n[0].view(np.uint32))   # the prompt ran un-sharded on rank 0: the single device's bits
        dev, clear = _tolerance_tier(lg_1, lg_n, ids_1, 0.25)
        if n == 4:
            lg_4 = lg_n
        print(f"{arch} wtype {wt}: {n} ranks behind one device: max|dlogit| {dev:.3e} sigma, steps with a clear margin {clear:.2f}")
    # FREE-RUNNING (each run feeds its own argmax back): the sharded steps are started AHEAD of the host from the third one on (decode-ahead, as on one device) -- the same bits
    # as with it switched off
    ids_a, lg_a, err_a = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"], threads=4, CLLM_HIP_TP="4", **turn2)
    ids_b, lg_b, err_b = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"], threads=4, CLLM_HIP_TP="4", CLLM_HIP_AHEAD="0", **turn2)
    assert sum("started ahead of the host" in ln for ln in err_a.splitlines()) >= 3 and "started ahead of the host" not in err_b, err_a[-1500:]
    assert ids_a == ids_b and np.array_equal(lg_a.view(np.uint32), lg_b.view(np.uint32))
    # the cross-stream path distinct GPUs take (every rank its own stream, the embedding row / position handed over behind an event, the ranks' streams joined into rank 0's at the
    # end of the step, gathers polling for scatters that run concurrently) on the one GPU: small shapes, every launch resident at once
    ids_s, lg_s, err = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"], teacher=ids_1, threads=4, CLLM_HIP_TP="2", CLLM_HIP_TP_STREAMS="1", CLLM_HIP_TP_DEBUG="1", **turn2)
    assert len([ln for ln in err.splitlines() if "-> tensor parallel over" in ln]) == n_dec - 1 and "timed out" not in err, err[-1500:]
    assert sum("replayed from the captured graphs" in ln or "started ahead of the host" in ln for ln in err.splitlines()) >= 4                 # one captured graph per stream
    ids_e, lg_e, err = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"], teacher=ids_1, threads=4, CLLM_HIP_TP="4", CLLM_HIP_TP_GRAPH="0", CLLM_HIP_TP_HEAD="0", **turn2)
    assert "replayed from the captured graphs" not in err and "lm_head rows sharded" not in err
    assert np.array_equal(lg_e.view(np.uint32), lg_4.view(np.uint32))                                         # replay and the sharded head change no bit of the sharded run
    dev, clear = _tolerance_tier(lg_1, lg_s, ids_1, 0.25)
    print(f"{arch} wtype {wt}: 2 ranks on streams of their own: max|dlogit| {dev:.3e} sigma")


@_REF_BUILT
def test_tensor_parallel_device_runs_everything_else_unsharded_and_bit_identical(gpu, tmp_path):
    """CLLM_HIP_TP=N changes nothing for graphs that are not the dense decode step: a sparse-MoE model (router, MUL_MAT_ID: tp_extract declines), a partial offload (`-ngl 1`: the
    scheduler splits the graph, no split is a whole step) and a flash-attention host (`-fa 1`: FLASH_ATTN_EXT instead of the fused attention block) all run un-sharded on rank 0 --
    ids and every logit word equal to the run without CLLM_HIP_TP"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config("tiny", max_len=64)
    prompt = [5, 9, 42, 300, 7, 99, 250]
    mx, ml = str(tmp_path / "mx.bin"), str(tmp_path / "ml.bin")
    make_ggmm.write_mixtral(mx, cfg, 12, seed=91)
    make_ggmm.write_model(ml, cfg, 12, seed=92)
    for mp, ngl, extra in ((mx, "all", {}), (ml, "1", {}), (ml, "all", dict(REF_CHAT_FA="1"))):
        ids_a, lg_a, _ = _host_run(tmp_path, mp, ngl, 10, prompt, cfg["vocab"], **extra)
        ids_b, lg_b, err = _host_run(tmp_path, mp, ngl, 10, prompt, cfg["vocab"], CLLM_HIP_TP="2", CLLM_HIP_TP_DEBUG="1", **extra)
        assert "-> tensor parallel over" not in err                              # no step was taken sharded ...
        assert ids_a == ids_b and np.array_equal(lg_a.view(np.uint32), lg_b.view(np.uint32)), (mp, ngl, extra)      # ... and nothing changed


@_BIG
@pytest.mark.parametrize("arch,cname,over,prompt_mod", [("qwen2", "qwen2-72b", dict(max_len=256, n_layer=2), 150000), ("llama3", "llama3-8b", dict(max_len=256, n_layer=2), 128000)])
def test_tensor_parallel_behind_the_boundary_at_real_block_shapes(gpu, tmp_path, arch, cname, over, prompt_mod):
    """BASELINE cfg4's and cfg2's BLOCK shapes under CLLM_HIP_TP=8 through the unmodified host (2 of the layers: 2.6 / 1.5 GB files): Qwen2-72B -- hidden 8192 (the gather prologue's
    four-values-per-lane x 4 passes form, 8 ranks' granules per element), 64 heads / 8 KV heads (one KV group per rank), ffn 29568 with the Q8_0 down projection (924 blocks ->
    4 x 116 + 4 x 115, gate/up rows follow), q/k/v biases sharded, a 152064-row lm_head sharded by rows; Llama-3-8B -- hidden 4096, ffn 14336 = 56 Q4_K blocks -> 7 per rank.
    Teacher-forced against the single-device run: tolerance tier; every step sharded and replayed from the captured graph."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ggmm
    cfg = gpu.synth.config(cname, **over)
    mp = str(tmp_path / "m.bin")
    make_ggmm.write_model(mp, cfg, 12, seed=41, fast=True, arch=arch)
    prompt = [(13 * i + 7) % prompt_mod for i in range(12)]
    n_dec = 10
    ids_1, lg_1, _ = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"], threads=8)
    ids_8, lg_8, err = _host_run(tmp_path, mp, "all", n_dec, prompt, cfg["vocab"], teacher=ids_1, threads=8, CLLM_HIP_TP="8", CLLM_HIP_TP_DEBUG="1")
    os.remove(mp)
    steps = [ln for ln in err.splitlines() if "-> tensor parallel over 8 ranks" in ln]
    assert len(steps) == n_dec and "timed out" not in err, err[-1500:]
    assert all("lm_head rows sharded" in ln for ln in steps) and sum("replayed from the captured graphs" in ln or "started ahead of the host" in ln for ln in steps) >= n_dec - 2
    assert np.array_equal(lg_1[0].view(np.uint32), lg_8[0].view(np.uint32))          # the prompt: un-sharded
    dev, clear = _tolerance_tier(lg_1, lg_8, ids_1, 0.25)
    print(f"{cname} block shapes, 8 ranks behind one device: max|dlogit| {dev:.3e} sigma, steps with a clear margin {clear:.2f}")


@_BIG
@pytest.mark.parametrize("wname,wt", [("q4_0", 2), ("q8_0", 8)])
def test_free_order_tier_at_llama3_8b_shapes_against_the_exact_order(gpu, wname, wt):
    """the review's contract question for the north star's other two weight types: the opt-in free-order decode (gemv_free32.hip) at BASELINE cfg2's shapes, FREE-RUNNING greedy
    from the same 16-token prompt against the exact-order default.  MEASURED (round 6): the 32 greedy ids are identical, the logits are NOT within the north star's 1e-3
    (max|dlogit| ~ 0.1 at sigma ~ 1: the Q8_0 activation quantizers of 32 layers turn last-bit differences of the fp32 fold into flipped int8 steps) -- which is why the exact
    order is the decode contract and this tier is opt-in only.  Asserted: what the tier does keep (ids over 32 steps, logits within 0.25 sigma), and that it is outside 1e-3."""
    sys.path.insert(0, ROOT)
    import bench
    L = gpu.lib.get()
    cfg = gpu.synth.config("llama3-8b", max_len=256)
    prompt = np.asarray(bench.PROMPT_IDS, np.int32)
    out = {}
    try:
        for mode in (0, 1):
            L.cllm_set_decode_free_order(mode)
            m = bench.build_model(gpu, cfg, wt, 0, 1)
            lg = [m.forward(prompt, n_past=0)]
            ids = []
            for _ in range(32):
                ids.append(int(np.argmax(lg[-1])))
                lg.append(m.decode_fused_logits(ids[-1]))
            out[mode] = (ids, np.stack(lg))
            m.close()
    finally:
        L.cllm_set_decode_free_order(0)
    (ids_e, lg_e), (ids_f, lg_f) = out[0], out[1]
    mism = sum(int(a != b) for a, b in zip(ids_e, ids_f))
    same = min([i for i, (a, b) in enumerate(zip(ids_e, ids_f)) if a != b] or [len(ids_e)])
    dev = float(np.max(np.abs(lg_e[:same + 1] - lg_f[:same + 1])))
    print(f"llama3-8b {wname}: free order vs exact order, free-running: id mismatches {mism}/32, max|dlogit| over the common prefix {dev:.3e} (sigma {float(np.std(lg_e)):.3f})")
    assert mism == 0 and dev < 0.25 * float(np.std(lg_e)), (mism, dev)
    assert dev > 1e-3                              # (the day this fails, the contract question is open again)
