#!/usr/bin/env python3
"""A/B timing of SpMV configurations in ONE process on ONE matrix allocation
(placement of the arrays in HBM changes SpMV time by several percent, so
configurations are compared on the same arrays: bis_mat_retune rebuilds the
row-block tables in place), interleaved rounds (cdna_hip_programming.md
section 5.4 rule 24).  Usage:
   python tools/spmv_ab.py hpcg 256 "variant=40,chunk=2048" "variant=40,chunk=1536,remap=1" ...
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import Context, load_library
import numpy as np

kind, n1 = sys.argv[1], int(sys.argv[2])
cfgs = [dict(kv.split("=") for kv in c.split(",")) for c in sys.argv[3:]]
lib = load_library()
ctx = Context(0)
A = ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_fem(n1) if kind == "fem" else ctx.gen_anderson(n1)
N = A.n_rows
x, y = ctx.alloc(N), ctx.alloc(N)
x.set(np.random.default_rng(0).uniform(-1, 1, N))
b = 12 * A.nnz + 20 * N
times = [[] for _ in cfgs]
for rnd in range(int(os.environ.get("ROUNDS", "5"))):
    for i, c in enumerate(cfgs):
        for k in ("variant", "chunk", "window", "xcd_remap", "packed", "lds_pad", "valdict"):
            lib.bis_set_option(("spmv_" + k).encode(), int(c.get(k if k != "xcd_remap" else "remap", -1)))
        ctx.check(lib.bis_mat_retune(ctx.h, A.h))
        for _ in range(2): ctx.spmv(A, x, y)
        ctx.sync(); ctx.profile(True)
        for _ in range(10): ctx.spmv(A, x, y)
        n, ms = ctx.profile_read(); ctx.profile(False)
        times[i].append(ms / n)
for i, c in enumerate(cfgs):
    t = np.array(times[i])
    print(f"{c}: median {np.median(t):.4f} ms min {t.min():.4f} max {t.max():.4f}  -> {b/np.median(t)/1e6:.0f} GB/s (median)", flush=True)
