#!/usr/bin/env python3
"""Per-block cycle stamps of the sliced-ELL SpMV (BIS_SELLWIN_DEBUG): where a workgroup's life goes.
   python tools/sellwin_debug.py [hpcg|anderson] [size]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
out = "/tmp/sellwin_dbg.txt"
os.environ["BIS_SELLWIN_DEBUG"] = out
from basic_iterative_solvers_amd import Context
kind = sys.argv[1] if len(sys.argv) > 1 else "hpcg"
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = Context(0)
A = ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_anderson(n1, shift=9.0)
x, y = ctx.alloc(A.n_rows), ctx.alloc(A.n_rows)
x.set(np.random.default_rng(0).uniform(-1, 1, A.n_rows))
for _ in range(5):
    ctx.spmv(A, x, y)
ctx.sync()
d = np.loadtxt(out)
t0 = d[:, 0] - d[:, 0].min()
print(f"{kind}-{n1}: {len(d)} blocks, stream info {A.spmv_stream_info()}")
print(f"kernel span {(t0 + d[:, 3]).max():.0f} cycles; per block (cycles): header arrived {d[:, 1].mean():.0f} (median {np.median(d[:, 1]):.0f}), "
      f"barrier passed {d[:, 2].mean():.0f} (median {np.median(d[:, 2]):.0f}), end {d[:, 3].mean():.0f} (median {np.median(d[:, 3]):.0f})")
print(f"compute part (end - barrier): mean {(d[:, 3] - d[:, 2]).mean():.0f}")
