#!/usr/bin/env python3
"""Which array's placement decides the win8 SpMV's time: (a) x and y fixed, the matrix (stream) re-made K times, every one
kept; (b) the matrix fixed, x and y re-allocated K times.   python tools/win8_place2.py [n] [K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from basic_iterative_solvers_amd import Context
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = Context()
ctx.set_option("spmv_valdict", 0)
N = n1 ** 3
xh = np.random.default_rng(1).uniform(-1, 1, N)


def t(A, x, y):
    for _ in range(3):
        ctx.spmv(A, x, y)
    ctx.sync(); ctx.profile(True)
    for _ in range(20):
        ctx.spmv(A, x, y)
    ctx.sync(); ctx.profile(False)
    n, ms = ctx.profile_read()
    return ms / n


x, y = ctx.upload(xh), ctx.alloc(N)
mats = []
for k in range(K):
    A = ctx.gen_hpcg(n1)
    mats.append(A)
    print(f"(a) matrix {k}, fixed x / y: {t(A, x, y):.4f} ms", flush=True)
vecs = []
for k in range(K):
    x2, y2 = ctx.upload(xh), ctx.alloc(N)
    vecs.append((x2, y2))
    print(f"(b) vectors {k}: " + " ".join(f"matrix {i}: {t(mats[i], x2, y2):.4f}" for i in range(min(K, 3))), flush=True)
ctx.close()
