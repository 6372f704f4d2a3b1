#!/usr/bin/env python3
"""Write a synthetic Llama-3-architecture model in chatllm.cpp's GGMM file format, with the SAME pre-quantized
blocks chatllm.cpp_amd/synth.py feeds to our own runner.  Used to drive the real reference host (oracle/_ref/main):
CPU backend vs `-ngl all` on our libggml-hip.so module with identical weights.

File layout (restated from SURVEY.md 8c; reader: /root/reference/src/models.cpp:1996-2047, src/chat.cpp:1425-1459):
  "ggmm", i32 version=1, u32 off_config, off_tokenizer, off_tensors, JSON meta padded to 4 bytes,
  @off_config   : i32 model_type (0x1700 = Llama3), i32 file version, BaseConfig (11 x i32), i32 num_kv_heads, f32 rope_theta
  @off_tokenizer: {i32 len, bytes, u8 type}* i32 -1 ; merges {i32 len, bytes}* i32 -1
  @off_tensors  : {i32 name_len, name, i32 ndim, i32 dims[ndim] (outer -> inner), i32 ggml_type, pad to 16, payload}*
"""
import argparse
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

WT = {"q4_k": 12, "q4_0": 2, "q4_1": 3, "q8_0": 8, "q5_k": 13, "q6_k": 14}
HF_NAME = {"wq": "self_attn.q_proj.weight", "wk": "self_attn.k_proj.weight", "wv": "self_attn.v_proj.weight", "wo": "self_attn.o_proj.weight",
           "wgate": "mlp.gate_proj.weight", "wup": "mlp.up_proj.weight", "wdown": "mlp.down_proj.weight",
           "attn_norm": "input_layernorm.weight", "ffn_norm": "post_attention_layernorm.weight"}


def bytes_to_unicode():
    bs = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, map(chr, cs)))


def write_model(path, cfg, wtype, seed=1234, model_name="SynthLlama3", fast=False, arch="llama3"):
    """arch "llama3" (MODEL_TYPE_LLAMA3 0x1700, models/llama.h:102-106) or "qwen2" (MODEL_TYPE_QWEN2 0x710, models/qwen.h:74-104: q/k/v biases,
    NEOX RoPE; the weights come from cfg with qkv_bias=1, rope_mode=2)"""
    pkg = ge.load_package()
    V, H = cfg["vocab"], cfg["hidden"]
    assert V >= 262
    b2u = bytes_to_unicode()
    toks = [(b2u[b].encode(), 1) for b in range(256)]
    if arch == "qwen2":
        assert cfg.get("qkv_bias") and cfg.get("rope_mode") == 2
        toks += [(s.encode(), 3) for s in ["<|endoftext|>", "<|im_start|>", "<|im_end|>"]]
    else:
        toks += [(s.encode(), 3) for s in ["<|begin_of_text|>", "<|end_of_text|>", "<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>"]]
    while len(toks) < V:
        toks.append((f"<|pad{len(toks)}|>".encode(), 3))
    with open(path, "wb") as f:
        f.write(b"ggmm")
        f.write(struct.pack("4i", 1, 0, 0, 0))
        meta = json.dumps({"model_name": model_name}).encode()
        f.write(meta + b"\0" * (-len(meta) % 4))

        def mark(off):
            p = f.tell()
            f.seek(off)
            f.write(struct.pack("i", p))
            f.seek(0, 2)
        mark(8)
        f.write(struct.pack("2i", 0x710 if arch == "qwen2" else 0x1700, 1))
        f.write(struct.pack("11i", wtype, V, H, cfg["n_head"], cfg["n_layer"], cfg["ffn"], cfg["max_len"], 256, 257, -1, -1))
        f.write(struct.pack("i", cfg["n_kv_head"]))
        if arch == "qwen2":
            f.write(struct.pack("i", cfg["max_len"]))               # sliding_window (not used below max_len)
        f.write(struct.pack("<f", cfg["rope_theta"]))
        mark(12)
        for t, tt in toks:
            f.write(struct.pack("i", len(t)))
            f.write(t)
            f.write(struct.pack("B", tt))
        f.write(struct.pack("i", -1))
        f.write(struct.pack("i", -1))
        mark(16)

        def dump(name, type_, dims_outer_to_inner, payload):
            nb = name.encode()
            f.write(struct.pack("i", len(nb)))
            f.write(nb)
            f.write(struct.pack("i", len(dims_outer_to_inner)))
            f.write(struct.pack(f"{len(dims_outer_to_inner)}i", *dims_outer_to_inner))
            f.write(struct.pack("i", type_))
            f.write(b"\0" * (-f.tell() % 16))
            f.write(np.ascontiguousarray(payload).tobytes())

        if fast:           # big models: the cheap generator bench.py uses (block bytes drawn directly)
            S = pkg.synth
            w = {name: (t, S.make_tensor_fast(name, t, rows, K, seed)) for name, t, rows, K in S.tensor_list(cfg, wtype)}
            w["out_norm"] = (0, S.make_norm("out_norm", H, seed))
            for i in range(cfg["n_layer"]):
                for k in ("attn_norm", "ffn_norm"):
                    w[f"layers.{i}.{k}"] = (0, S.make_norm(f"layers.{i}.{k}", H, seed))
                if cfg.get("qkv_bias"):
                    hd = cfg["head_dim"]
                    for b, nb_ in (("bq", cfg["n_head"] * hd), ("bk", cfg["n_kv_head"] * hd), ("bv", cfg["n_kv_head"] * hd)):
                        w[f"layers.{i}.{b}"] = (0, S.make_bias(f"layers.{i}.{b}", nb_, seed))
        else:
            w = pkg.synth.make_model(cfg, wtype, seed=seed)
        shape = {n: (rows, K) for n, _, rows, K in pkg.synth.tensor_list(cfg, wtype)}
        dump("model.embed_tokens.weight", w["tok_embd"][0], [V, H], w["tok_embd"][1])
        for i in range(cfg["n_layer"]):
            p, hp = f"layers.{i}.", f"model.layers.{i}."
            for k in ("attn_norm", "wdown", "wgate", "wup", "ffn_norm", "wk", "wo", "wq", "wv"):
                t, arr = w[p + k]
                dims = [H] if k.endswith("norm") else list(shape[p + k])
                dump(hp + HF_NAME[k], t, dims, arr)
            if arch == "qwen2":
                for b, nm in (("bq", "q_proj"), ("bk", "k_proj"), ("bv", "v_proj")):
                    dump(hp + f"self_attn.{nm}.bias", 0, [len(w[p + b][1])], w[p + b][1])
        dump("model.norm.weight", 0, [H], w["out_norm"][1])
        dump("lm_head.weight", w["lm_head"][0], [V, H], w["lm_head"][1])
    return path


def write_mixtral(path, cfg, wtype, seed=1234, n_expert=8, n_used=2, fast=False):
    """a Mixtral-architecture model (MODEL_TYPE_MIXTRAL 0x601, models/mistral.h:44-170): 8 experts / top 2 as the reference's template
    requires, sliding window 4096, llama-v2 style vocabulary records {i32 len, bytes, f32 score} (src/tokenizer.cpp:310-372, 430-441);
    expert weights as individual tensors block_sparse_moe.experts.E.w1/w2/w3 (gate/down/up), router block_sparse_moe.gate"""
    pkg = ge.load_package()
    S = pkg.synth
    V, H, F, hd = cfg["vocab"], cfg["hidden"], cfg["ffn"], cfg["head_dim"]
    QD, KD = cfg["n_head"] * hd, cfg["n_kv_head"] * hd
    assert QD == H, "the reference derives head_dim from hidden / heads for this architecture"
    with open(path, "wb") as f:
        f.write(b"ggmm")
        f.write(struct.pack("4i", 1, 0, 0, 0))
        meta = json.dumps({"model_name": "SynthMixtral"}).encode()
        f.write(meta + b"\0" * (-len(meta) % 4))

        def mark(off):
            p = f.tell()
            f.seek(off)
            f.write(struct.pack("i", p))
            f.seek(0, 2)
        mark(8)
        f.write(struct.pack("2i", 0x601, 1))
        f.write(struct.pack("11i", wtype, V, H, cfg["n_head"], cfg["n_layer"], F, cfg["max_len"], 1, 2, -1, -1))
        f.write(struct.pack("2i", cfg["n_kv_head"], 4096))          # num_key_value_heads, sliding_window
        f.write(struct.pack("<f", cfg["rope_theta"]))
        f.write(struct.pack("2i", n_used, n_expert))                # num_experts_per_tok, num_local_experts
        mark(12)
        toks = [b"<unk>", b"<s>", b"</s>"] + [f"<0x{b:02X}>".encode() for b in range(256)]
        while len(toks) < V:
            toks.append(f"t{len(toks)}".encode())
        for t in toks[:V]:
            f.write(struct.pack("i", len(t)))
            f.write(t)
            f.write(struct.pack("<f", 0.0))
        f.write(struct.pack("i", -1))
        mark(16)

        def dump(name, type_, dims_outer_to_inner, payload):
            nb = name.encode()
            f.write(struct.pack("i", len(nb)))
            f.write(nb)
            f.write(struct.pack("i", len(dims_outer_to_inner)))
            f.write(struct.pack(f"{len(dims_outer_to_inner)}i", *dims_outer_to_inner))
            f.write(struct.pack("i", type_))
            f.write(b"\0" * (-f.tell() % 16))
            f.write(np.ascontiguousarray(payload).tobytes())

        gen = S.make_tensor_fast if fast else S.make_tensor        # fast: real-shape models (block bytes drawn directly)

        mix = cfg.get("mix") or {}                                  # per-tensor types by name fragment (Q4_K_M-style files keep expert down projections in Q6_K): {".w2.": 14}

        def q(name, rows, K):
            t = next((v for k, v in mix.items() if k in name), wtype)
            dump("model." + name if not name.startswith("lm_head") else name, t, [rows, K], gen("mixtral." + name, t, rows, K, seed))

        q("embed_tokens.weight", V, H)
        for i in range(cfg["n_layer"]):
            hp = f"layers.{i}."
            dump("model." + hp + "input_layernorm.weight", 0, [H], S.make_norm("mixtral." + hp + "attn_norm", H, seed))
            for e in range(n_expert):
                q(hp + f"block_sparse_moe.experts.{e}.w1.weight", F, H)
                q(hp + f"block_sparse_moe.experts.{e}.w2.weight", H, F)
                q(hp + f"block_sparse_moe.experts.{e}.w3.weight", F, H)
            q(hp + "block_sparse_moe.gate.weight", n_expert, H)
            dump("model." + hp + "post_attention_layernorm.weight", 0, [H], S.make_norm("mixtral." + hp + "ffn_norm", H, seed))
            q(hp + "self_attn.k_proj.weight", KD, H)
            q(hp + "self_attn.o_proj.weight", H, QD)
            q(hp + "self_attn.q_proj.weight", QD, H)
            q(hp + "self_attn.v_proj.weight", KD, H)
        dump("model.norm.weight", 0, [H], S.make_norm("mixtral.out_norm", H, seed))
        q("lm_head.weight", V, H)
    return path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="tiny")
    ap.add_argument("--wtype", default="q4_k", choices=sorted(WT))
    ap.add_argument("--max-len", type=int, default=256)
    ap.add_argument("--out", required=True)
    ap.add_argument("--fast", action="store_true")
    ap.add_argument("--arch", default="llama3", choices=["llama3", "mixtral", "qwen2"])
    ap.add_argument("--layers", type=int, default=0, help="override the number of blocks (profiling a few real-shape layers)")
    a = ap.parse_args()
    pkg = ge.load_package()
    over = dict(qkv_bias=1, rope_mode=2, rope_theta=1e6) if a.arch == "qwen2" else {}
    if a.layers: over["n_layer"] = a.layers
    cfg = pkg.synth.config(a.config, max_len=a.max_len, **over)
    if a.arch == "mixtral":
        write_mixtral(a.out, cfg, WT[a.wtype], fast=a.fast)
    else:
        write_model(a.out, cfg, WT[a.wtype], fast=a.fast, arch=a.arch)
    print(a.out, os.path.getsize(a.out), "bytes")
