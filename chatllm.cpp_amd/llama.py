"""Python handle on the C++ decoder runner (csrc/decoder.hip, cllm_llama_* in include/chatllm_hip.h):
the host-side mirror of the reference's generate loop for Llama-3 / Qwen2 style models
(BaseModelForConditionalGeneration::generate_next_token, src/models.cpp:1108-1123)."""
import ctypes as C

import numpy as np

from . import lib as _l


class Llama:
    def __init__(self, cfg, weights=None, tp_rank=0, tp_size=1, ffn_local=0):
        _l.require_gpu()
        self.cfg = dict(cfg)
        c = _l.LlamaConfig(cfg["n_layer"], cfg["hidden"], cfg["n_head"], cfg["n_kv_head"], cfg["head_dim"], cfg["ffn"],
                           cfg["vocab"], cfg["max_len"], cfg.get("rope_mode", 0), cfg.get("rope_theta", 500000.0),
                           cfg.get("rms_eps", 1e-5), 1 if cfg.get("qkv_bias") else 0, tp_rank, tp_size, ffn_local)
        self.h = C.c_void_p()
        _l.check(_l.get().cllm_llama_create(C.byref(c), None, C.byref(self.h)), "llama_create")
        self.n_past = 0
        self._cb = None
        if weights:
            for name, (t, arr) in weights.items():
                self.set_weight(name, t, arr)

    def set_weight(self, name, type_, arr):
        arr = np.ascontiguousarray(arr)
        _l.check(_l.get().cllm_llama_set_weight(self.h, name.encode(), type_, arr.ctypes.data_as(C.c_void_p), arr.nbytes),
                 f"set_weight({name})")

    def bind_weight(self, name, type_, dev_ptr, nbytes):
        _l.check(_l.get().cllm_llama_bind_weight(self.h, name.encode(), type_, C.c_void_p(dev_ptr), nbytes), f"bind_weight({name})")

    def set_allreduce(self, fn):
        """fn(stream_ptr, buf_ptr, n_floats): sum `buf` over the tensor-parallel group on `stream`"""
        self._cb = _l.ALLREDUCE_FN(lambda user, stream, buf, n: fn(stream, buf, n))
        _l.check(_l.get().cllm_llama_set_allreduce(self.h, self._cb, None), "set_allreduce")

    def set_tp_comm(self, comm):
        """bind an RCCL communicator (lib cllm_tp_init): all-reduces run on the runner's stream, inside the decode graph"""
        _l.check(_l.get().cllm_llama_set_tp_comm(self.h, comm), "set_tp_comm")

    def set_tp_oneshot(self, handle):
        """bind a one-shot all-reduce group (lib cllm_tp_oneshot_create / _connect): one kernel launch per all-reduce, inside the decode graph"""
        _l.check(_l.get().cllm_llama_set_tp_oneshot(self.h, handle), "set_tp_oneshot")

    def set_tp_fused(self, handle):
        """bind the receive buffers of the fused all-reduce (lib cllm_tp_fused_create / _connect): the decode steps run without all-reduce launches"""
        _l.check(_l.get().cllm_llama_set_tp_fused(self.h, handle), "set_tp_fused")

    def use_graph(self, enable):
        _l.check(_l.get().cllm_llama_use_graph(self.h, 1 if enable else 0), "use_graph")

    def forward(self, tokens, n_past=None):
        """run the tokens at positions n_past.., return logits[vocab] (host) of the last one"""
        tokens = np.ascontiguousarray(tokens, np.int32)
        if n_past is None:
            n_past = self.n_past
        logits = np.zeros(self.cfg["vocab"], np.float32)
        _l.check(_l.get().cllm_llama_forward(self.h, tokens.ctypes.data_as(C.c_void_p), tokens.size, n_past, None,
                                             logits.ctypes.data_as(C.c_void_p)), "llama_forward")
        self.n_past = n_past + tokens.size
        return logits

    def decode_greedy(self, first_token, n_steps, n_past=None):
        """n_steps greedy steps starting from `first_token` at position n_past; returns the generated ids"""
        if n_past is None:
            n_past = self.n_past
        out = np.zeros(n_steps, np.int32)
        _l.check(_l.get().cllm_llama_decode_greedy(self.h, int(first_token), n_past, n_steps, out.ctypes.data_as(C.c_void_p)),
                 "decode_greedy")
        self.n_past = n_past + n_steps
        return out

    def decode_fused_logits(self, token, n_past=None):
        """logits of `token` at position n_past through the fused single-token kernels (no sampling)"""
        if n_past is None:
            n_past = self.n_past
        logits = np.zeros(self.cfg["vocab"], np.float32)
        _l.check(_l.get().cllm_llama_decode_fused_logits(self.h, int(token), n_past, logits.ctypes.data_as(C.c_void_p)), "decode_fused_logits")
        self.n_past = n_past + 1
        return logits

    def debug_read(self, what, n):
        out = np.zeros(n, np.float32)
        _l.check(_l.get().cllm_llama_debug_read(self.h, what.encode(), out.ctypes.data_as(C.c_void_p), n), "debug_read")
        return out

    def weight_bytes(self):
        return _l.get().cllm_llama_weight_bytes(self.h)

    def close(self):
        if self.h:
            _l.get().cllm_llama_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
