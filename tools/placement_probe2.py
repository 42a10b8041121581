#!/usr/bin/env python3
"""Is "slow placement" visible to a pure streaming read?  For each of K copies of HPCG-256: SpMV ms and
the GB/s of a streaming norm over the copy's val array (3.6 GB), plus the address of val."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from basic_iterative_solvers_amd import Context, Vec, load_library
K = int(sys.argv[1]) if len(sys.argv) > 1 else 36
lib = load_library()
ctx = Context(0)
mats = [ctx.gen_hpcg(256) for _ in range(K)]
N = mats[0].n_rows
x, y = ctx.alloc(N), ctx.alloc(N)
x.set(np.random.default_rng(0).uniform(-1, 1, N))
def ptrs(A):
    a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
    lib.bis_mat_debug_ptrs(A.h, C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value
for i, A in enumerate(mats):
    rp, col, val = ptrs(A)
    v = Vec(ctx, val, A.nnz, False)
    for _ in range(2): ctx.spmv(A, x, y)
    ctx.sync(); ctx.profile(True)
    for _ in range(8): ctx.spmv(A, x, y)
    n, ms = ctx.profile_read(); ctx.profile(False)
    import time
    ctx.euclidean_vec_norm(v); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3): ctx.euclidean_vec_norm(v)
    ctx.sync(); dt = (time.perf_counter() - t0) / 3
    print(f"copy {i:2d}: spmv {ms / n:.4f} ms   stream-read of val {8 * A.nnz / dt / 1e9:7.0f} GB/s   val @ {val:#x} col @ {col:#x}", flush=True)
print(f"x @ {x.ptr:#x}  y @ {y.ptr:#x}")
