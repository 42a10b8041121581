#!/usr/bin/env python3
"""Per-tile cycle stamps of the tiled sweep (BIS_TRSV_TILE_DEBUG): where a tile's time goes.
   python tools/trsv_tile_debug.py anderson 256 [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = "/tmp/tile_dbg.bin"
os.environ["BIS_TRSV_TILE_DEBUG"] = F
import numpy as np
from basic_iterative_solvers_amd import Context
kind, n1 = sys.argv[1], int(sys.argv[2])
ctx = Context(0)
if len(sys.argv) > 3: ctx.set_option("trsv_tile_rows", int(sys.argv[3]))
if len(sys.argv) > 4: ctx.set_option("trsv_tile_edge", int(sys.argv[4]))
ctx.set_option("trsv_tiled", int(os.environ.get("TILED_MODE", "-1")))
A = ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_fem(n1) if kind == "fem" else ctx.gen_anderson(n1, shift=9.0)
Ls, Us, D, Dinv = ctx.split_strict(A)
N = A.n_rows
b, x = ctx.alloc(N), ctx.alloc(N)
ctx.init_vector(b, 1.0)
for _ in range(2):
    ctx.sptrsv(Ls, x, D, b); ctx.sync()
d = np.fromfile(F, dtype=np.int64).reshape(-1, 8)
t0 = d[:, 0].min()
f = 100.0  # s_memrealtime ticks per us (100 MHz); the wait counters are core cycles (~2400 per us)
start, end = (d[:, 0] - t0) / f, (d[:, 1] - t0) / f
print(f"tiles {len(d)}  sweep {end.max():.1f} us (first start -> last end)")
dur = end - start
steps, loop_cyc = d[:, 7] & 0xffff, d[:, 7] >> 16
print(f"tile duration us: mean {dur.mean():.1f} median {np.median(dur):.1f} max {dur.max():.1f}; steps/tile mean {steps.mean():.1f}")
work = loop_cyc - d[:, 2] - d[:, 3]
print(f"step loop: mean {loop_cyc.mean()/2400:.1f} us/tile, of it neither waiting for loaders nor for operands {work.mean()/2400:.1f} us = {work.sum()/steps.sum():.0f} core cycles per step")
print(f"compute wave waited for loaders: mean {d[:,2].mean()/2400:.1f} us/tile; for external operands: mean {d[:,3].mean()/2400:.1f} us/tile")
for nm, c in (("entry loader", 4), ("slot loader", 5), ("poller", 6)):
    print(f"{nm} finished after (from tile start) mean {((d[:,c]-d[:,0])/f).mean():.1f} us")
k = np.arange(len(d))
for i in (0, 1, 2, 3, len(d)//2, len(d)//2+1, len(d)-1):
    print(f"tile {i}: start {start[i]:.1f} end {end[i]:.1f} wait_load {d[i,2]/2400:.1f} wait_ext {d[i,3]/2400:.1f}")
conc = [(np.sum((start <= t) & (end > t))) for t in np.linspace(0, end.max(), 21)[1:-1]]
print("tiles in flight at 5%..95% of the sweep:", conc)
