#!/usr/bin/env python3
"""SpMV time over the life of a process (chunks of 10 launches): clock ramp / throttling phases?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from basic_iterative_solvers_amd import Context
ctx = Context(0)
A = ctx.gen_hpcg(256)
N = A.n_rows
x, y = ctx.alloc(N), ctx.alloc(N)
x.set(np.random.default_rng(0).uniform(-1, 1, N))
t_start = time.perf_counter()
series = []
for chunk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    ctx.profile(True)
    for _ in range(10): ctx.spmv(A, x, y)
    n, ms = ctx.profile_read(); ctx.profile(False)
    series.append((time.perf_counter() - t_start, ms / n))
s = np.array(series)
for i in range(0, len(s), 10):
    seg = s[i:i + 10]
    print(f"t={seg[0,0]:6.2f}s  " + " ".join(f"{v:.3f}" for v in seg[:, 1]), flush=True)
