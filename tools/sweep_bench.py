#!/usr/bin/env python3
"""Time the natural-order sweeps of a matrix WITHOUT a grid hint under several kernel choices, in one process, and compare the
results bit for bit (first configuration = reference).
   python tools/sweep_bench.py unstr:80,80,80 rcm  chain=1 "chain=0,wave=1" "chain=0,wave=0"
   python tools/sweep_bench.py fem:80,80,81 asis  "tiled=0,chain=1" "tiled=0,chain=0" tiled=-1
order: asis | rcm | bfs (device ordering, bis_mat_bfs_order + bis_mat_permute)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import Context, load_library
import numpy as np

spec, order = sys.argv[1], sys.argv[2]
cfgs = [dict(kv.split("=") for kv in c.split(",")) for c in sys.argv[3:]]
lib = load_library()
ctx = Context(0)
kind, rest = spec.split(":")
nums = [int(v) for v in rest.split(",")]
t0 = time.perf_counter()
A = {"unstr": ctx.gen_unstr, "fem": ctx.gen_fem, "hpcg": ctx.gen_hpcg}[kind](*nums) if kind != "anderson" else ctx.gen_anderson(nums[0], shift=9.0)
if order != "asis":
    perm = ctx.bfs_order(A, rcm=(order == "rcm"))
    B = ctx.permute(A, perm)
    A.free()
    A = B
ctx.sync()
print(f"{spec} {order}: {A.n_rows} rows, {A.nnz} non-zeros, set up in {time.perf_counter() - t0:.2f} s", flush=True)
N = A.n_rows
b, x = ctx.alloc(N), ctx.alloc(N)
b.set(np.random.default_rng(1).uniform(-1, 1, N))
ref = None
for c in cfgs:
    for k in ("tiled", "chain", "wave", "grid", "batch", "chain_idle", "chain_pause", "chain_pairs", "chain_prefix"):
        lib.bis_set_option(("trsv_" + k).encode(), int(c.get(k, -1)))
    Ls, Us, D, Dinv = ctx.split_strict(A)  # plans are cached per matrix: fresh triangles per configuration
    t0 = time.perf_counter(); ctx.sptrsv(Ls, x, D, b); ctx.sync(); t_first = time.perf_counter() - t0
    fw = x.to_host()
    ctx.bsptrsv(Us, x, D, b); ctx.sync()
    bw = x.to_host()
    if ref is None:
        ref = (fw, bw)
    same = bool(np.array_equal(fw, ref[0]) and np.array_equal(bw, ref[1]))
    ts = []
    for solve, T in ((ctx.sptrsv, Ls), (ctx.bsptrsv, Us)):
        solve(T, x, D, b); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(10): solve(T, x, D, b)
        ctx.sync(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"  {c}: forward {ts[0]:.3f} ms  backward {ts[1]:.3f} ms  bit-identical to the first configuration: {same}  "
          f"(first forward sweep incl. plan {t_first:.2f} s)", flush=True)
    for m in (Ls, Us):
        m.free()
    D.free(); Dinv.free()
