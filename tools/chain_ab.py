#!/usr/bin/env python3
"""Same box, same call: the chained sweep of two builds of the library (tools/ab_libs/*.so) on the config-5 inputs.
    python tools/chain_ab.py <lib.so> [prefix option]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import basic_iterative_solvers_amd as B
B.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
ctx = B.Context()
if len(sys.argv) > 2:
    ctx.set_option("trsv_chain_prefix", int(sys.argv[2]))
ctx.set_option("trsv_tiled", 0)
for rep in range(2):
    r = bench.sweep_legs(ctx, sweeps=10, warm=3, only=("unstr80_rcm", "fem80x80x81"))
    for k, v in r.items():
        print(os.path.basename(sys.argv[1]), sys.argv[2:], k, {d: round(v[d]["avg_sweep_ms"], 3) for d in ("forward", "backward")}, flush=True)
ctx.close()
