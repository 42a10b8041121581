#!/usr/bin/env python3
"""Per-chunk SpMV time inside the fused CG loop over the life of a process, then stand-alone SpMVs on the
same matrix/vectors, then CG again."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from basic_iterative_solvers_amd import Context
ctx = Context(0)
A = ctx.gen_hpcg(256)
N = A.n_rows
b, x = ctx.alloc(N), ctx.alloc(N)
ctx.init_vector(b, 1.0); ctx.init_vector(x, 0.1)
cg = ctx.cg(A, b, x, None)
cg.init(0.0)
def cg_series(chunks, label):
    out = []
    for c in range(chunks):
        ctx.sync(); t0 = time.perf_counter()
        ctx.profile(True)
        cg.iterate(10)
        ctx.sync(); dt = (time.perf_counter() - t0) / 10
        n, ms = ctx.profile_read(); ctx.profile(False)
        out.append((ms / n, dt * 1e3))
    print(label, "spmv+dot ms per chunk:", " ".join(f"{a:.3f}" for a, _ in out), flush=True)
    print(label, "iteration ms per chunk:", " ".join(f"{d:.3f}" for _, d in out), flush=True)
def spmv_series(chunks, label):
    u, v = ctx.alloc(N), ctx.alloc(N)
    ctx.init_vector(u, 0.5)
    out = []
    for c in range(chunks):
        ctx.profile(True)
        for _ in range(10): ctx.spmv(A, u, v)
        n, ms = ctx.profile_read(); ctx.profile(False)
        out.append(ms / n)
    print(label, "plain spmv ms per chunk:", " ".join(f"{a:.3f}" for a in out), flush=True)
cg_series(20, "CG#1")
spmv_series(12, "SPMV")
cg_series(12, "CG#2")
