#!/usr/bin/env python3
"""A/B of SpMV settings INSIDE the fused CG iteration (SpMV with the (Ap,p)
epilogue), one matrix allocation, interleaved rounds.
   python tools/cg_ab.py hpcg 256 "chunk=1024" "chunk=2048" ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import Context, load_library
import numpy as np
kind, n1 = sys.argv[1], int(sys.argv[2])
cfgs = [dict(kv.split("=") for kv in c.split(",")) for c in sys.argv[3:]]
lib = load_library(); ctx = Context(0)
A = ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_anderson(n1, shift=9.0)
N = A.n_rows
b, x = ctx.alloc(N), ctx.alloc(N)
res = [[] for _ in cfgs]
for rnd in range(int(os.environ.get("ROUNDS", "4"))):
    for i, c in enumerate(cfgs):
        for k in ("variant", "chunk_fused", "window", "xcd_remap", "packed"):
            lib.bis_set_option(("spmv_" + k).encode(), int(c.get({"xcd_remap": "remap", "chunk_fused": "chunk"}.get(k, k), -1)))
        lib.bis_set_option(b"cg_nt_x", int(c.get("ntx", -1)))
        lib.bis_set_option(b"spmv_valdict", int(c.get("vd", -1)))
        lib.bis_set_option(b"spmv_win8_depth", int(c.get("w8d", -1)))
        lib.bis_set_option(b"spmv_sellwin_nt", int(c.get("ntc", -1)))
        ctx.check(lib.bis_mat_retune(ctx.h, A.h))
        ctx.init_vector(b, 1.0); ctx.init_vector(x, 0.1)
        cg = ctx.cg(A, b, x); cg.init(0.0); cg.iterate(5); ctx.sync()
        ctx.profile(True); t0 = time.perf_counter(); cg.iterate(30); ctx.sync(); t1 = time.perf_counter()
        n, ms = ctx.profile_read(); ctx.profile(False); cg.free()
        res[i].append((ms / n, (t1 - t0) / 30 * 1e3))
for i, c in enumerate(cfgs):
    r = np.array(res[i])
    print(f"{c}: spmv+dot {np.median(r[:,0]):.4f} ms  iteration {np.median(r[:,1]):.4f} ms", flush=True)
