"""Device tensor = {type, ne, nb, data}: the same four fields as the reference's ggml_tensor
(ggml/include/ggml.h:656-688), with view/permute/reshape helpers that only touch metadata
(chatllm::ggml::view_*/permute/reshape, src/layers.cpp)."""
import ctypes as C

import numpy as np

from . import lib as _l

F32, F16, Q4_0, Q4_1, Q8_0, Q4_K, Q5_K, Q6_K, I32, I64 = 0, 1, 2, 3, 8, 12, 13, 14, 26, 27
Q5_0, Q5_1, Q2_K, Q3_K, IQ4_NL, MXFP4, IQ4_XS, TQ1_0, TQ2_0, IQ2_XXS, IQ2_XS, IQ3_XXS, IQ3_S, IQ2_S, IQ1_S, IQ1_M = 6, 7, 10, 11, 20, 39, 23, 34, 35, 16, 17, 18, 21, 22, 19, 29          # mat-mul (any columns) and GET_ROWS
TYPE_SIZE = {F32: 4, F16: 2, Q4_0: 18, Q4_1: 20, Q8_0: 34, Q4_K: 144, Q5_K: 176, Q6_K: 210, I32: 4, I64: 8, Q5_0: 22, Q5_1: 24, Q2_K: 84, Q3_K: 110, IQ4_NL: 18, MXFP4: 17, IQ4_XS: 136, TQ1_0: 54, TQ2_0: 66, IQ2_XXS: 66, IQ2_XS: 74, IQ2_S: 82, IQ3_XXS: 98, IQ3_S: 110, IQ1_S: 50, IQ1_M: 56}
BLCK = {F32: 1, F16: 1, Q4_0: 32, Q4_1: 32, Q8_0: 32, Q4_K: 256, Q5_K: 256, Q6_K: 256, I32: 1, I64: 1, Q5_0: 32, Q5_1: 32, Q2_K: 256, Q3_K: 256, IQ4_NL: 32, MXFP4: 32, IQ4_XS: 256, TQ1_0: 256, TQ2_0: 256, IQ2_XXS: 256, IQ2_XS: 256, IQ2_S: 256, IQ3_XXS: 256, IQ3_S: 256, IQ1_S: 256, IQ1_M: 256}
NP_OF = {F32: np.float32, F16: np.float16, I32: np.int32, I64: np.int64}


def row_size(t, ne):
    assert ne % BLCK[t] == 0, "row length must be a multiple of the block size"
    return TYPE_SIZE[t] * (ne // BLCK[t])


class Buffer:
    """an owned device allocation (buffer_i.free_buffer on garbage collection)"""

    def __init__(self, nbytes):
        _l.require_gpu()
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        _l.check(_l.get().cllm_malloc(C.byref(self.ptr), C.c_size_t(max(self.nbytes, 16))), "cllm_malloc")

    def __del__(self):
        try:
            if getattr(self, "ptr", None) is not None and self.ptr.value:
                _l.get().cllm_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


class Tensor:
    def __init__(self, type_, ne, nb=None, buf=None, offset=0):
        ne = [int(x) for x in ne] + [1] * (4 - len(ne))
        if nb is None:
            nb = [TYPE_SIZE[type_], row_size(type_, ne[0])]
            nb.append(nb[1] * ne[1])
            nb.append(nb[2] * ne[2])
        self.type, self.ne, self.nb = type_, ne, [int(x) for x in nb]
        if buf is None:
            buf = Buffer(self.nb[3] * ne[3])
        self.buf, self.offset = buf, int(offset)

    # ---- host <-> device (buffer_i.set_tensor / get_tensor) ----
    @staticmethod
    def from_numpy(arr, type_=None, ne=None):
        """arr: numpy array. float/int arrays map to F32/F16/I32/I64 with ne = reversed shape;
        quantized tensors take a uint8 array of raw blocks plus explicit type_ and ne."""
        arr = np.ascontiguousarray(arr)
        if type_ is None:
            type_ = {np.dtype(np.float32): F32, np.dtype(np.float16): F16, np.dtype(np.int32): I32, np.dtype(np.int64): I64}[arr.dtype]
        if ne is None:
            ne = list(reversed(arr.shape))
        t = Tensor(type_, ne)
        assert arr.nbytes == t.nbytes(), f"byte size mismatch: {arr.nbytes} vs {t.nbytes()}"
        if arr.nbytes:
            _l.check(_l.get().cllm_memcpy_h2d(t.data_ptr(), arr.ctypes.data_as(C.c_void_p), arr.nbytes, None), "h2d")
            _l.check(_l.get().cllm_stream_sync(None), "sync")
        return t

    def nbytes(self):
        return self.nb[3] * self.ne[3]

    def data_ptr(self):
        return C.c_void_p(self.buf.ptr.value + self.offset)

    def is_contiguous(self):
        nb = TYPE_SIZE[self.type]
        if self.nb[0] != nb:
            return False
        nb = row_size(self.type, self.ne[0])
        for i in range(1, 4):
            if self.ne[i] != 1 and self.nb[i] != nb:
                return False
            nb *= self.ne[i]
        return True

    def raw(self):
        """the whole underlying buffer as bytes (for byte-exact comparisons of caches)"""
        out = np.zeros(self.buf.nbytes, np.uint8)
        _l.check(_l.get().cllm_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.buf.ptr, self.buf.nbytes, None), "d2h")
        return out

    def numpy(self):
        """dense copy on the host; quantized tensors come back as raw block bytes"""
        assert self.is_contiguous(), "numpy(): make it contiguous with ops.cont first"
        n = self.nbytes()
        out = np.zeros(n, np.uint8)
        if n:
            _l.check(_l.get().cllm_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.data_ptr(), n, None), "d2h")
        if self.type in NP_OF:
            shape = list(reversed(self.ne))
            while len(shape) > 1 and shape[0] == 1:      # drop the unused outer dimensions (ne[3], ne[2], ...)
                shape.pop(0)
            return out.view(NP_OF[self.type]).reshape(shape)
        return out

    def c(self):
        t = _l.CTensor()
        t.type = self.type
        t.ne[:] = self.ne
        t.nb[:] = self.nb
        t.data = self.data_ptr().value
        return t

    # ---- metadata-only views ----
    def view(self, ne, nb, offset=0):
        ne = list(ne) + [1] * (4 - len(ne))
        nb = list(nb)
        while len(nb) < 4:
            nb.append(nb[-1] * ne[len(nb) - 1])
        return Tensor(self.type, ne, nb, self.buf, self.offset + offset)

    def reshape(self, *ne):
        assert self.is_contiguous()
        assert int(np.prod(ne)) == int(np.prod(self.ne))
        return Tensor(self.type, ne, None, self.buf, self.offset)

    def permute(self, a0, a1, a2, a3):
        """ggml_permute: dimension i of self becomes dimension a_i of the result"""
        ne, nb = [0] * 4, [0] * 4
        for i, a in enumerate((a0, a1, a2, a3)):
            ne[a], nb[a] = self.ne[i], self.nb[i]
        return Tensor(self.type, ne, nb, self.buf, self.offset)

    def transpose(self):
        return self.permute(1, 0, 2, 3)
