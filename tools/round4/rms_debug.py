#!/usr/bin/env python3
"""round 4 debug: where does the GPU RMS_NORM differ from the oracle?  (words differing per row, the scale each side used)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); O = ge.load_oracle(); pkg.lib.require_gpu()
rng = np.random.default_rng(11)
for n0 in (8, 100, 4096, 8192, 8192, 16384, 12288, 5120, 1000):
    bad = 0
    for rep in range(40):
        rows = 1 + rep % 3
        x = rng.standard_normal((rows, n0)).astype(np.float32)
        want = np.zeros_like(x)
        O.rms_norm(O.tensor(x, O.F32, [n0, rows]), O.tensor(want, O.F32, [n0, rows]), 1e-5)
        got = pkg.ops.rms_norm(pkg.Tensor.from_numpy(x), 1e-5).numpy().reshape(x.shape)
        if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
            bad += 1
            for r in range(rows):
                d = got[r].view(np.uint32) != want[r].view(np.uint32)
                if d.any():
                    i = int(np.argmax(d))
                    print(f"n0={n0} rows={rows} row {r}: {int(d.sum())} of {n0} words differ; first at {i}: x={x[r,i]!r} got={got[r,i]!r} want={want[r,i]!r}  ratio got/x={got[r,i]/x[r,i]!r} want/x={want[r,i]/x[r,i]!r}")
    print(f"n0={n0}: {bad} of 40 calls differ")
