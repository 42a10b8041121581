#!/usr/bin/env python3
"""Is the win8 SpMV's time a matter of WHEN (clock ramp of a freshly started process) or WHERE (placement of the arrays)?
One allocation, batches of 20 launches back to back for a few seconds; then a second allocation, the same.
    python tools/win8_timeline.py [n] [batches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from basic_iterative_solvers_amd import Context
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = Context()
ctx.set_option("spmv_valdict", 0)
N = n1 ** 3
t00 = time.perf_counter()
held = []
for alloc in range(3):
    A = ctx.gen_hpcg(n1)
    x, y = ctx.upload(np.random.default_rng(1).uniform(-1, 1, N)), ctx.alloc(N)
    ctx.spmv(A, x, y); ctx.sync()
    out = []
    for b in range(B):
        ctx.profile(True)
        for _ in range(20):
            ctx.spmv(A, x, y)
        ctx.sync(); ctx.profile(False)
        n, ms = ctx.profile_read()
        out.append((time.perf_counter() - t00, ms / n))
    print(f"allocation {alloc}: " + " ".join(f"{t:.2f}s:{v:.3f}" for t, v in out[::max(1, B // 12)]), flush=True)
    held.append((A, x, y))
ctx.close()
