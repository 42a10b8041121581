#!/usr/bin/env python3
"""Probe (GPU box): SpMV bandwidth vs row length / structure on host-built
banded matrices (checks e.g. LDS bank conflicts of the row-sum phase for even
row lengths, and irregular rows)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from basic_iterative_solvers_amd import Context
from oracle.pyoracle import CRS, Oracle
ctx = Context(0); orc = Oracle()
n = 1 << 20
rng = np.random.default_rng(5)
def banded(L, spread):
    # L entries per row at pseudo-random offsets within +-spread (sorted, distinct), clipped
    offs = np.sort(rng.choice(np.arange(-spread, spread + 1), size=L, replace=False))
    rows = np.arange(n)[:, None]
    cols = np.clip(rows + offs[None, :], 0, n - 1).astype(np.int32)
    rp = np.arange(n + 1, dtype=np.int64) * L
    return CRS(n, rp, cols.ravel(), rng.uniform(-1, 1, n * L))
def ragged(mean):
    lens = rng.integers(1, 2 * mean, n)
    rp = np.concatenate([[0], np.cumsum(lens)])
    cols = (np.repeat(np.arange(n), lens) + rng.integers(-2000, 2000, rp[-1])).clip(0, n - 1).astype(np.int32)
    return CRS(n, rp, cols, rng.uniform(-1, 1, rp[-1]))
x = rng.uniform(-1, 1, n)
dx, dy = ctx.upload(x), ctx.alloc(n)
cases = [(f"banded L={L} spread={sp}", banded(L, sp)) for L, sp in [(7, 300), (8, 300), (16, 300), (27, 300), (32, 300), (64, 300), (73, 300), (73, 20000)]]
cases.append(("ragged mean 30", ragged(30)))
for name, A in cases:
    dA = ctx.matrix(A)
    ctx.spmv(dA, dx, dy); y = dy.to_host()
    yo = orc.spmv(A, x)
    err = np.max(np.abs(y - yo)) / np.max(np.abs(yo))
    ctx.sync(); ctx.profile(True)
    for _ in range(10): ctx.spmv(dA, dx, dy)
    k, ms = ctx.profile_read(); ctx.profile(False)
    b = 12 * A.nnz + 20 * n
    print(f"{name:28s}: {ms/k:.4f} ms  {b/(ms/k)/1e6:.0f} GB/s  relerr {err:.1e}", flush=True)
    dA.free()
