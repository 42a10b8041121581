#!/usr/bin/env python3
"""One workload of bench.py's `sweeps` legs, one direction, N sweeps and nothing else after the setup -- the program the
rocprofv3 passes of tools/trsv_traffic.sh profile.
    python tools/sweep_probe.py <hpcg256|anderson256|fem80x80x81|unstr80_asis|unstr80_rcm> <forward|backward> [sweeps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from basic_iterative_solvers_amd import Context  # noqa: E402

key, direction = sys.argv[1], sys.argv[2]
n_sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
case = next(c for c in bench.SWEEP_CASES if c[0] == key)
ctx = Context()
A = case[2](ctx)
if case[3]:
    B = ctx.permute(A, ctx.bfs_order(A, rcm=True))
    A.free()
    A = B
N = A.n_rows
Ls, Us, D, Dinv = ctx.split_strict(A)
b, x = ctx.upload(np.random.default_rng(21).uniform(-1, 1, N)), ctx.alloc(N)
T, solve = (Us, ctx.bsptrsv) if direction == "backward" else (Ls, ctx.sptrsv)
ctx.profile(True)
for _ in range(n_sweeps):
    solve(T, x, D, b)
ctx.sync()
n, ms = ctx.profile_read_sweeps()
print(f"{key} {direction}: {n} sweeps, last-{n - 1} average {ms / n:.3f} ms (first one builds the plan), kernel {T.sweep_kernel(direction == 'backward')}, "
      f"nnz_T {T.nnz}, rows {N}")
ctx.close()
