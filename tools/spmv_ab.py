#!/usr/bin/env python3
"""A/B timing of SpMV configurations in ONE process, interleaved rounds
(cdna_hip_programming.md section 5.4 rule 24).  Usage:
   python tools/spmv_ab.py hpcg 256 "variant=40,chunk=2048,window=0" "variant=40,chunk=1536,window=0" ...
"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import Context, load_library
import numpy as np

kind, n1 = sys.argv[1], int(sys.argv[2])
cfgs = [dict(kv.split("=") for kv in c.split(",")) for c in sys.argv[3:]]
lib = load_library()
ctx = Context(0)
mats = []
for c in cfgs:
    lib.bis_set_option(b"spmv_variant", int(c.get("variant", -1)))
    lib.bis_set_option(b"spmv_chunk", int(c.get("chunk", -1)))
    lib.bis_set_option(b"spmv_window", int(c.get("window", -1)))
    mats.append(ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_anderson(n1))
N = mats[0].n_rows
x, y = ctx.alloc(N), ctx.alloc(N)
x.set(np.random.default_rng(0).uniform(-1, 1, N))
b = 12 * mats[0].nnz + 20 * N
times = [[] for _ in cfgs]
for rnd in range(int(os.environ.get("ROUNDS", "8"))):
    for i, c in enumerate(cfgs):
        lib.bis_set_option(b"spmv_variant", int(c.get("variant", -1)))
        lib.bis_set_option(b"spmv_window", int(c.get("window", -1)))
        for _ in range(2): ctx.spmv(mats[i], x, y)
        ctx.sync(); ctx.profile(True)
        for _ in range(10): ctx.spmv(mats[i], x, y)
        n, ms = ctx.profile_read(); ctx.profile(False)
        times[i].append(ms / n)
for i, c in enumerate(cfgs):
    t = np.array(times[i])
    print(f"{c}: median {np.median(t):.4f} ms min {t.min():.4f} max {t.max():.4f}  -> {b/np.median(t)/1e6:.0f} GB/s (median)", flush=True)
