#!/usr/bin/env python3
"""Per-iteration time of the DISTRIBUTED CG code path on one GPU (world = 1, RCCL
self-communicator): the launch/collective overhead of one rank's share of a
strong-scaled problem, without the wire.  nz = 256/N emulates rank work at N GPUs.
   python tools/dist_overhead.py 256 32 [iters]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from basic_iterative_solvers_amd import Context, Dist, rccl_unique_id

n1, nz = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
ctx = Context(0)
for mode in ("single", "dist-rccl"):
    A = ctx.gen_hpcg(n1, n1, nz)
    N = A.n_rows
    if mode == "single":
        b, x = ctx.alloc(N), ctx.alloc(N)
        ctx.init_vector(b, 1.0); ctx.init_vector(x, 0.1)
        cg = ctx.cg(A, b, x)
    else:
        d = Dist(ctx, A, 0, 1, np.array([0, N], dtype=np.int64))
        d.set_send_lists(np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int32))
        d.use_rccl(rccl_unique_id(ctx))
        b, x = ctx.alloc(d.n_local), ctx.alloc(d.n_local)
        ctx.init_vector(b, 1.0); ctx.init_vector(x, 0.1)
        cg = d.cg(b, x)
    cg.init(0.0)
    cg.iterate(20); ctx.sync()
    t0 = time.perf_counter()
    cg.iterate(iters); 
    t_enq = time.perf_counter() - t0
    ctx.sync()
    t1 = time.perf_counter() - t0
    print(f"{mode}: rows {N}  {1e3 * t1 / iters:.4f} ms/iteration (host enqueue {1e3 * t_enq / iters:.4f} ms/iteration)", flush=True)
