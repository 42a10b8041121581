#!/usr/bin/env python3
"""Tuning (GPU box): time bis_sptrsv on the strict lower triangle of a
generated matrix for several persistent-grid sizes."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
from basic_iterative_solvers_amd import Context
kind, n1 = sys.argv[1], int(sys.argv[2])
ctx = Context(0)
A = ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_anderson(n1, shift=9.0)
Ls, Us, D, Dinv = ctx.split_strict(A)
N = A.n_rows
b, x = ctx.alloc(N), ctx.alloc(N)
ctx.init_vector(b, 1.0)
t0 = time.time(); ctx.sptrsv(Ls, x, D, b); ctx.sync(); t_first = time.time() - t0
reps = 5
t0 = time.time()
for _ in range(reps): ctx.sptrsv(Ls, x, D, b)
ctx.sync(); dt = (time.time() - t0) / reps
print(json.dumps(dict(first_s=t_first, ms=dt * 1e3, nnz=Ls.nnz)))
'''
kind = sys.argv[1]; n1 = sys.argv[2]
for g in [int(v) for v in os.environ.get("GRIDS", "0,8,16,32,64,128,256,512,1024").split(",")]:
    env = dict(os.environ)
    if g: env["BIS_TRSV_GRID"] = str(g)
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT, kind, n1], env=env, capture_output=True, text=True, timeout=300)
    try:
        r = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{kind}-{n1} grid {g or 'auto':>5}: {r['ms']:.3f} ms/solve (first incl. analysis {r['first_s']:.2f} s)", flush=True)
    except Exception:
        print("FAILED", g, out.stderr[-500:], flush=True)
