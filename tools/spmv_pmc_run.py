#!/usr/bin/env python3
"""Run N SpMV launches of one configuration (for rocprofv3 --pmc passes).
   python tools/spmv_pmc_run.py hpcg 256 "packed=1,variant=20,chunk=2010,remap=8" [launches]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import Context, load_library
import numpy as np

kind, n1 = sys.argv[1], int(sys.argv[2])
cfg = dict(kv.split("=") for kv in sys.argv[3].split(",")) if len(sys.argv) > 3 and sys.argv[3] else {}
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 5
lib = load_library()
for k in ("variant", "chunk", "window", "xcd_remap", "packed", "lds_pad"):
    lib.bis_set_option(("spmv_" + k).encode(), int(cfg.get(k if k != "xcd_remap" else "remap", -1)))
ctx = Context(0)
A = ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_anderson(n1)
N = A.n_rows
x, y = ctx.alloc(N), ctx.alloc(N)
x.set(np.random.default_rng(0).uniform(-1, 1, N))
for _ in range(launches):
    ctx.spmv(A, x, y)
ctx.sync()
print("done", cfg, flush=True)
