#!/usr/bin/env python3
"""K matrices (each kept: the next lands elsewhere), 10 win8 launches each, placement search off: the program the counter passes of
tools/win8_place_pmc.sh profile to see what differs between a slow and a fast allocation of the stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from basic_iterative_solvers_amd import Context
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = Context()
ctx.set_option("spmv_valdict", 0)
ctx.set_option("spmv_win8_tune", 0)
N = 256 ** 3
x, y = ctx.upload(np.random.default_rng(1).uniform(-1, 1, N)), ctx.alloc(N)
mats = []
for k in range(K):
    A = ctx.gen_hpcg(256)
    mats.append(A)
    ctx.spmv(A, x, y); ctx.sync()
    ctx.profile(True)
    for _ in range(10):
        ctx.spmv(A, x, y)
    ctx.sync(); ctx.profile(False)
    n, ms = ctx.profile_read()
    print(f"matrix {k}: {ms / n:.4f} ms", flush=True)
ctx.close()
