#!/usr/bin/env python3
"""How much of the win8 SpMV's time is the PLACEMENT of its arrays: the same matrix generated K times in one process (every
generation allocates anew), HIP-event timed, with the addresses.  python tools/win8_place.py [n] [K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from basic_iterative_solvers_amd import Context
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = Context()
ctx.set_option("spmv_valdict", 0)
N = n1 ** 3
keep = []
for k in range(K):
    A = ctx.gen_hpcg(n1)
    x, y = ctx.upload(np.random.default_rng(1).uniform(-1, 1, N)), ctx.alloc(N)
    for _ in range(3):
        ctx.spmv(A, x, y)
    ctx.sync(); ctx.profile(True)
    for _ in range(20):
        ctx.spmv(A, x, y)
    ctx.sync(); ctx.profile(False)
    n, ms = ctx.profile_read()
    print(f"trial {k}: {ms / n:.4f} ms  x {x.ptr:#x} y {y.ptr:#x}", flush=True)
    if k % 2 == 0:
        keep.append((A, x, y))   # keep every other allocation alive: the next ones land elsewhere
    else:
        A.free(); x.free(); y.free()
ctx.close()
