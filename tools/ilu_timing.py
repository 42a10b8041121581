#!/usr/bin/env python3
"""Setup-time breakdown on the config-5 stand-in: python tools/ilu_timing.py fem 80"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import Context
kind, n1 = sys.argv[1], int(sys.argv[2])
ctx = Context(0)
def T(label, f):
    ctx.sync(); t0 = time.perf_counter(); r = f(); ctx.sync()
    print(f"{label}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True); return r
A = T("generate", lambda: ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_fem(n1) if kind == "fem" else ctx.gen_anderson(n1, shift=9.0))
print("rows", A.n_rows, "nnz", A.nnz)
sp = T("split_strict", lambda: ctx.split_strict(A))
for o in sp: o.free()
Ls, L_D, Us, U_D = T("ilu0 (1st)", lambda: ctx.ilu0(A))
if A.nnz < 2**31:
    for o in T("ilu0 (2nd)", lambda: ctx.ilu0(A)): o.free()
N = A.n_rows
b, x = ctx.alloc(N), ctx.alloc(N)
ctx.init_vector(b, 1.0)
T("sptrsv L first (plan)", lambda: ctx.sptrsv(Ls, x, L_D, b))
T("sptrsv L second", lambda: ctx.sptrsv(Ls, x, L_D, b))
T("bsptrsv U first (plan)", lambda: ctx.bsptrsv(Us, x, U_D, b))
T("bsptrsv U second", lambda: ctx.bsptrsv(Us, x, U_D, b))
