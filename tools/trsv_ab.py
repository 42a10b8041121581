#!/usr/bin/env python3
"""Time bis_sptrsv / bis_bsptrsv (natural order) for several option sets in one process, and compare the results bit for bit.
   python tools/trsv_ab.py hpcg 256 fresh=1 edge=264200 wgs=2 backoff=0          (tiled sweep: tile edges ex | ey << 8 | ez << 16,
                                                                                 workgroups per CU, poller back-off; fresh=1: new plan)
   python tools/trsv_ab.py hpcg 256 fresh=1 exp=128 exp=16                        (timing-only builds, TiledArgs::exp_flags: results wrong)
   python tools/trsv_ab.py hpcg 128 tiled=0 "tiled=0,one_xcd=1" "tiled=0,grid=64" (level-scheduled kernels)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import Context, load_library
import numpy as np
kind, n1 = sys.argv[1], int(sys.argv[2])
cfgs = [dict(kv.split("=") for kv in c.split(",")) for c in sys.argv[3:]]
lib = load_library()
ctx = Context(0)
A = ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_fem(n1) if kind == "fem" else ctx.gen_anderson(n1, shift=9.0)
Ls, Us, D, Dinv = ctx.split_strict(A)
N = A.n_rows
b, x, ref = ctx.alloc(N), ctx.alloc(N), None
b.set(np.random.default_rng(1).uniform(-1, 1, N))
for c in cfgs:
    lib.bis_set_option(b"trsv_tiled", int(c.get("tiled", -1)))
    lib.bis_set_option(b"trsv_tile_rows", int(c.get("rows", -1)))
    lib.bis_set_option(b"trsv_tile_wgs", int(c.get("wgs", -1)))
    lib.bis_set_option(b"trsv_tile_edge", int(c.get("edge", -1)))
    lib.bis_set_option(b"trsv_tile_exp", int(c.get("exp", -1)))
    lib.bis_set_option(b"trsv_tile_backoff", int(c.get("backoff", -1)))
    if "rows" in c or "fresh" in c or "edge" in c:  # plans are cached per matrix: a new tile size needs fresh triangles
        Ls, Us, D, Dinv = ctx.split_strict(A)
    t0 = time.perf_counter(); ctx.sptrsv(Ls, x, D, b); ctx.sync(); t_first = time.perf_counter() - t0
    lib.bis_set_option(b"trsv_one_xcd", int(c.get("one_xcd", -1)))
    lib.bis_set_option(b"trsv_grid", int(c.get("grid", -1)))
    lib.bis_set_option(b"trsv_batch", int(c.get("batch", -1)))
    lib.bis_set_option(b"trsv_by_pos", int(c.get("pos", -1)))
    lib.bis_set_option(b"trsv_wave", int(c.get("wave", -1)))
    ctx.sptrsv(Ls, x, D, b); ctx.sync()
    got = x.to_host()
    if ref is None: ref = got
    same = bool(np.array_equal(got, ref))
    ts = []
    for solve, T in ((ctx.sptrsv, Ls), (ctx.bsptrsv, Us)):
        solve(T, x, D, b); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5): solve(T, x, D, b)
        ctx.sync(); ts.append((time.perf_counter() - t0) / 5 * 1e3)
    print(f"{kind}-{n1} {c}: forward {ts[0]:.3f} ms  backward {ts[1]:.3f} ms  bit-identical to first config: {same}  (first forward solve incl. plan {t_first:.2f} s)", flush=True)
