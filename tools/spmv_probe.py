#!/usr/bin/env python3
"""Run N SpMVs (and N fused-dot SpMVs through a CG object) on one generated matrix: the program the SpMV counter
passes profile.  python3 tools/spmv_probe.py hpcg 256 20 [valdict]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import Context, load_library
import numpy as np
kind, n1, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = load_library()
if len(sys.argv) > 4:
    lib.bis_set_option(b"spmv_valdict", int(sys.argv[4]))
ctx = Context(0)
if kind == "unstr_rcm":  # config 5 as a real mesh is multiplied
    A0 = ctx.gen_unstr(n1)
    A = ctx.permute(A0, ctx.bfs_order(A0, rcm=True))
    A0.free()
elif kind == "unstr_asis":  # config 5 as generated: no locality (column slabs; BIS_SPMV_COLSLAB=0 / 6 pick the one pass / six slabs without a trial)
    A = ctx.gen_unstr(n1)
else:
    A = ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_fem(n1) if kind == "fem" else ctx.gen_anderson(n1)
x, y = ctx.alloc(A.n_rows), ctx.alloc(A.n_rows)
x.set(np.random.default_rng(0).uniform(-1, 1, A.n_rows))
for _ in range(reps):
    ctx.spmv(A, x, y)
ctx.sync()
print("done", A.n_rows, A.nnz)
