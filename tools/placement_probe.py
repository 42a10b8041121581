#!/usr/bin/env python3
"""Does the SpMV time depend on WHICH allocation holds the matrix?  Several copies of HPCG-256 live in one
process; time the SpMV on each (same x, y), interleaved.   python tools/placement_probe.py [copies]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from basic_iterative_solvers_amd import Context
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = Context(0)
mats = [ctx.gen_hpcg(256) for _ in range(copies)]
N = mats[0].n_rows
x, y = ctx.alloc(N), ctx.alloc(N)
x.set(np.random.default_rng(0).uniform(-1, 1, N))
times = [[] for _ in mats]
for rnd in range(4):
    for i, A in enumerate(mats):
        for _ in range(2): ctx.spmv(A, x, y)
        ctx.sync(); ctx.profile(True)
        for _ in range(10): ctx.spmv(A, x, y)
        n, ms = ctx.profile_read(); ctx.profile(False)
        times[i].append(ms / n)
for i, t in enumerate(times):
    print(f"copy {i}: median {np.median(t):.4f} ms  (min {min(t):.4f}, max {max(t):.4f})", flush=True)
# and the vectors: re-allocate x, y and repeat on copy 0
for k in range(3):
    x2, y2 = ctx.alloc(N), ctx.alloc(N)
    x2.set(np.random.default_rng(0).uniform(-1, 1, N))
    for _ in range(2): ctx.spmv(mats[0], x2, y2)
    ctx.sync(); ctx.profile(True)
    for _ in range(10): ctx.spmv(mats[0], x2, y2)
    n, ms = ctx.profile_read(); ctx.profile(False)
    print(f"copy 0 with fresh vectors #{k}: {ms / n:.4f} ms", flush=True)
