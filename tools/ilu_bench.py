#!/usr/bin/env python3
"""Device ILU(0): one persistent launch against a launch per level, same input, factors compared bit for bit.
   python tools/ilu_bench.py fem:80,80,81 asis | unstr:80,80,80 rcm"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import Context, load_library
import numpy as np
spec, order = sys.argv[1], sys.argv[2]
lib = load_library()
ctx = Context(0)
kind, rest = spec.split(":")
nums = [int(v) for v in rest.split(",")]
A = {"unstr": ctx.gen_unstr, "fem": ctx.gen_fem, "hpcg": ctx.gen_hpcg}[kind](*nums)
if order != "asis":
    perm = ctx.bfs_order(A, rcm=(order == "rcm"))
    B = ctx.permute(A, perm); A.free(); A = B
ctx.sync()
print(f"{spec} {order}: {A.n_rows} rows, {A.nnz} non-zeros", flush=True)
ref = None
for mode in (1, 0, 1):
    lib.bis_set_option(b"ilu0_persistent", mode)
    ctx.sync(); t0 = time.perf_counter()
    Ls, L_D, Us, U_D = ctx.ilu0(A)
    ctx.sync(); dt = time.perf_counter() - t0
    got = (Ls.download()[2], Us.download()[2], U_D.to_host())
    if ref is None: ref = got
    same = all(np.array_equal(a, b) for a, b in zip(got, ref))
    print(f"  ilu0_persistent={mode}: {dt * 1e3:.1f} ms (whole call: sorted copy, level analysis, factorisation, split), factors identical to the first run: {same}", flush=True)
    for o in (Ls, Us): o.free()
