#!/usr/bin/env python3
"""What one row of a chain costs the chained sweep (bis_trsv_chain.hip): strictly lower BAND matrices (row r has the w rows before
it as operands) are one long dependency chain -- the sweep is n in-chain hops plus a hand-off through memory every 128 rows.
   python tools/chain_probe.py [n] [w ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import Context, load_library
from oracle.pyoracle import CRS
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
widths = [int(v) for v in sys.argv[2:]] or [1, 3, 13, 35, 64]
lib = load_library()
ctx = Context(0)
rng = np.random.default_rng(3)
for w in widths:
    lens = np.minimum(np.arange(n), w)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    col = np.concatenate([np.arange(r - lens[r], r) for r in range(n)]).astype(np.int32)
    val = rng.uniform(-1, 1, rp[-1]) / max(w, 1)
    L = ctx.matrix(CRS(n, rp, col, val))
    D, b, x = ctx.upload(np.full(n, 2.0)), ctx.upload(rng.uniform(-1, 1, n)), ctx.alloc(n)
    out = []
    for chain in (1, 0):
        lib.bis_set_option(b"trsv_chain", chain)
        Lc = ctx.matrix(CRS(n, rp, col, val))
        ctx.sptrsv(Lc, x, D, b); ctx.sync()
        ref = x.to_host()
        t0 = time.perf_counter()
        for _ in range(3): ctx.sptrsv(Lc, x, D, b)
        ctx.sync()
        out.append(((time.perf_counter() - t0) / 3, ref))
        Lc.free()
    same = np.array_equal(out[0][1], out[1][1])
    print(f"band w={w:3d}, {n} rows: chained {out[0][0] * 1e3:8.3f} ms = {out[0][0] / n * 1e6:6.3f} us per row;  level-scheduled {out[1][0] * 1e3:8.3f} ms = "
          f"{out[1][0] / n * 1e6:6.3f} us per row;  identical: {same}", flush=True)
    L.free()
