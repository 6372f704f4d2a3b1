"""The C-ABI entries round 6 added for the ggml module (include/chatllm_hip.h), each called directly and checked against numpy on the same inputs: the decode-ahead's one-launch
prep (snapshot + greedy sampler + scalar updates), the KV-cache shard copies and the pitched copy of the logical tensor-parallel device, and the fused all-reduce's scatter /
gather forms over an in-process rank group (byte / index work and rank-ordered fp32 sums: bit-exact)."""
import ctypes as C

import numpy as np
import pytest

from synth_helpers import rand_blocks

pytestmark = pytest.mark.gpu
rng = np.random.default_rng(66)


def _dev(gpu, arr):
    """a device buffer holding arr's bytes"""
    arr = np.ascontiguousarray(arr)
    b = gpu.tensor.Buffer(max(arr.nbytes, 16))
    L = gpu.lib.get()
    gpu.lib.check(L.cllm_memcpy_h2d(b.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes, None), "h2d")
    gpu.lib.check(L.cllm_stream_sync(None), "sync")
    return b


def _host(gpu, buf, nbytes, dtype):
    out = np.zeros(nbytes, np.uint8)
    gpu.lib.check(gpu.lib.get().cllm_memcpy_d2h(out.ctypes.data_as(C.c_void_p), buf.ptr, nbytes, None), "d2h")
    return out.view(dtype)


@pytest.mark.parametrize("n", [2048, 128256, 152064 + 3])
def test_snapshot_argmax_set_is_copy_first_maximum_and_stores(gpu, n):
    """cllm_op_snapshot_argmax_set: the ranges are copied byte for byte (16-byte and 4-byte paths), the token is std::max_element's FIRST maximum (Sampler greedy,
    src/models.cpp:676-690) also when the maximum repeats across workgroup slices, every (pointer, value) record is stored, and the launch can be repeated (ticket reset)"""
    L = gpu.lib.get()
    n4 = n - n % 4 if n % 4 else n
    for rep in range(3):
        x = rng.standard_normal(n4).astype(np.float32)
        top = float(x.max()) + 1.0
        pos = sorted(rng.choice(n4, 5, replace=False).tolist())
        x[pos] = top                                           # five equal maxima: the first one wins
        if rep == 2:
            x[:] = -np.inf; x[n4 - 1] = -1e30                   # everything else -inf
            pos = [n4 - 1]
        dx = _dev(gpu, x)
        other = rng.integers(0, 2**31, 1027, dtype=np.int32)  # 4108 bytes: the 4-byte path
        dother = _dev(gpu, other)
        snap = gpu.tensor.Buffer(n4 * 4 + 8192)
        ranges = np.zeros(2 * 3, np.uint64)
        ranges[0:3] = (dx.ptr.value, snap.ptr.value, n4 * 4)
        ranges[3:6] = (dother.ptr.value, snap.ptr.value + n4 * 4 + 64, other.nbytes)
        dr = _dev(gpu, ranges)
        scal = _dev(gpu, np.zeros(8, np.int32))
        recs = np.zeros(5 * 2, np.uint64)                       # { int32 * ptr; int32 val; int32 pad }
        for i in range(5):
            recs[2 * i] = scal.ptr.value + 4 * (i + 1)
            recs[2 * i + 1] = np.uint64(1000 + 7 * i + rep)
        dt = _dev(gpu, recs)
        scratch = _dev(gpu, np.zeros(1024, np.int32))
        tok_host = gpu.tensor.Buffer(16)                        # (device memory stands in for the page-locked word)
        for again in range(2):
            gpu.lib.check(L.cllm_op_snapshot_argmax_set(None, dr.ptr, 2, dx.ptr, n4, scal.ptr, tok_host.ptr, dt.ptr, 5, scratch.ptr), "snapshot_argmax_set")
            gpu.lib.check(L.cllm_stream_sync(None), "sync")
            s = _host(gpu, scal, 32, np.int32)
            assert s[0] == pos[0] and _host(gpu, tok_host, 4, np.int32)[0] == pos[0]
            assert list(s[1:6]) == [1000 + 7 * i + rep for i in range(5)]
            got = _host(gpu, snap, n4 * 4 + 64 + other.nbytes, np.uint8)
            assert np.array_equal(got[: n4 * 4].view(np.float32).view(np.uint32), x.view(np.uint32))
            assert np.array_equal(got[n4 * 4 + 64:].view(np.int32), other)


@pytest.mark.parametrize("kd_full,kd_shard,off,ML", [(1024, 128, 256, 96), (256, 64, 192, 40), (512, 512, 0, 33)])
def test_kv_shard_copy_moves_exactly_the_rows_and_columns_asked_for(gpu, kd_full, kd_shard, off, ML):
    """cllm_op_kv_shard_copy both ways: rows [p0, p1) of columns [off, off + kd_shard) between the host's caches (K [n][kd_full], V [kd_full][ML]) and dense shards; the
    one-row form reads the position on the device; nothing outside the slice changes"""
    L = gpu.lib.get()
    nl = 3
    K = rng.integers(0, 65536, (nl, ML, kd_full), dtype=np.uint16); V = rng.integers(0, 65536, (nl, kd_full, ML), dtype=np.uint16)
    sk = rng.integers(0, 65536, (nl, ML, kd_shard), dtype=np.uint16); sv = rng.integers(0, 65536, (nl, kd_shard, ML), dtype=np.uint16)
    dK = [_dev(gpu, K[l]) for l in range(nl)]; dV = [_dev(gpu, V[l]) for l in range(nl)]
    dsk = [_dev(gpu, sk[l]) for l in range(nl)]; dsv = [_dev(gpu, sv[l]) for l in range(nl)]
    tab = np.array([[dK[l].ptr.value, dV[l].ptr.value, dsk[l].ptr.value, dsv[l].ptr.value] for l in range(nl)], np.uint64)
    dtab = _dev(gpu, tab)
    p0, p1 = 5, ML - 7
    gpu.lib.check(L.cllm_op_kv_shard_copy(None, dtab.ptr, nl, kd_shard, kd_full, off, ML, p0, p1, None, 0), "to shard")
    gpu.lib.check(L.cllm_stream_sync(None), "sync")
    esk, esv = sk.copy(), sv.copy()
    esk[:, p0:p1, :] = K[:, p0:p1, off:off + kd_shard]
    esv[:, :, p0:p1] = V[:, off:off + kd_shard, p0:p1]
    for l in range(nl):
        assert np.array_equal(_host(gpu, dsk[l], sk[l].nbytes, np.uint16).reshape(ML, kd_shard), esk[l])
        assert np.array_equal(_host(gpu, dsv[l], sv[l].nbytes, np.uint16).reshape(kd_shard, ML), esv[l])
        assert np.array_equal(_host(gpu, dK[l], K[l].nbytes, np.uint16).reshape(ML, kd_full), K[l])          # the source is untouched
    # one row back, the position on the device
    pos = ML - 3
    dpos = _dev(gpu, np.array([pos], np.int32))
    gpu.lib.check(L.cllm_op_kv_shard_copy(None, dtab.ptr, nl, kd_shard, kd_full, off, ML, 0, 0, dpos.ptr, 1), "to host")
    gpu.lib.check(L.cllm_stream_sync(None), "sync")
    eK, eV = K.copy(), V.copy()
    eK[:, pos, off:off + kd_shard] = esk[:, pos, :]
    eV[:, off:off + kd_shard, pos] = esv[:, :, pos]
    for l in range(nl):
        assert np.array_equal(_host(gpu, dK[l], K[l].nbytes, np.uint16).reshape(ML, kd_full), eK[l])
        assert np.array_equal(_host(gpu, dV[l], V[l].nbytes, np.uint16).reshape(kd_full, ML), eV[l])
    assert L.cllm_op_kv_shard_copy(None, dtab.ptr, nl, kd_shard, kd_full, kd_full - kd_shard + 1, ML, 0, 1, None, 0) != 0     # the slice must fit the rows


def test_copy_2d_cuts_whole_quant_blocks_out_of_every_row(gpu):
    """cllm_copy_2d as the K-split of a quantized matrix: blocks [b0, b1) of every row of a Q8_0 [K, N] matrix"""
    L = gpu.lib.get()
    N, K, b0, b1 = 37, 32 * 19, 5, 12
    w = rand_blocks(8, N, K, rng)                               # uint8 [N, 19 * 34]
    dw = _dev(gpu, w)
    out = gpu.tensor.Buffer(N * (b1 - b0) * 34)
    gpu.lib.check(L.cllm_copy_2d(None, out.ptr, (b1 - b0) * 34, C.c_void_p(dw.ptr.value + b0 * 34), 19 * 34, (b1 - b0) * 34, N), "copy_2d")
    gpu.lib.check(L.cllm_stream_sync(None), "sync")
    assert np.array_equal(_host(gpu, out, N * (b1 - b0) * 34, np.uint8).reshape(N, -1), w[:, b0 * 34:b1 * 34])


@pytest.mark.parametrize("wtype,nranks", [(12, 2), (12, 8), (8, 4), (2, 3)])
def test_in_process_rank_group_scatter_then_gather_is_the_rank_ordered_sum(gpu, wtype, nranks):
    """cllm_tp_fused_create_group + cllm_op_mul_mat_vec_tp_scatter + cllm_op_tp_gather_residual on one stream (virtual ranks): every rank's partial rows are the bits of the plain
    mat-vec over its K-shard, and the gathered residual is  x + (((p0 + p1) + p2) + ...)  in fp32, rank order, identical on every rank; a second step reuses the slots"""
    L = gpu.lib.get()
    T = gpu.Tensor
    H, blk = 1024, (256 if wtype == 12 else 32)
    Kr = [blk * (2 + r % 2) for r in range(nranks)]            # uneven K-shards (whole blocks)
    devs = (C.c_int * nranks)(*([0] * nranks))
    objs = (C.c_void_p * nranks)()
    gpu.lib.check(L.cllm_tp_fused_create_group(nranks, devs, 3, H, objs), "create_group")
    try:
        ws = [T.from_numpy(rand_blocks(wtype, H, Kr[r], rng), wtype, [Kr[r], H]) for r in range(nranks)]
        cws = [w.c() for w in ws]
        for step in range(2):
            xs = [T.from_numpy(rng.standard_normal((1, Kr[r])).astype(np.float32)) for r in range(nranks)]
            resid = rng.standard_normal(H).astype(np.float32)
            dres = T.from_numpy(resid.reshape(1, H))
            parts = []
            for r in range(nranks):
                y = T(gpu.F32, [H, 1])
                gpu.lib.check(L.cllm_op_mul_mat_vec_fused(None, C.byref(cws[r]), 2, xs[r].data_ptr(), None, 0.0, 0, None, y.data_ptr()), "plain")
                parts.append(y.numpy().reshape(H).copy())
            site = 1 + step
            for r in range(nranks):
                gpu.lib.check(L.cllm_tp_fused_advance(objs[r], None), "advance")
            for r in range(nranks):
                gpu.lib.check(L.cllm_op_mul_mat_vec_tp_scatter(None, C.byref(cws[r]), 2, xs[r].data_ptr(), objs[r], site), "scatter")
            want = parts[0].copy()
            for r in range(1, nranks):
                want = want + parts[r]
            want = resid + want
            for r in range(nranks):
                out = T(gpu.F32, [H, 1])
                gpu.lib.check(L.cllm_op_tp_gather_residual(None, dres.data_ptr(), H, objs[r], site, out.data_ptr()), "gather")
                gpu.lib.check(L.cllm_stream_sync(None), "sync")
                assert np.array_equal(out.numpy().reshape(H).view(np.uint32), want.view(np.uint32)), (r, step)
                assert L.cllm_tp_fused_error(objs[r]) == 0
    finally:
        for r in range(nranks):
            L.cllm_tp_fused_destroy(objs[r])
