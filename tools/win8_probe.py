#!/usr/bin/env python3
"""Time the SpMV kernels that stream 8-byte values on one matrix: the row-block kernel on the CRS arrays (spmv_win8 = 0) and
the window + sliced-ELL form (win8) at several block sizes / ring depths.  HIP events on the library's stream.
    python tools/win8_probe.py <hpcg:256 | anderson:256 | fem:80,80,81 | unstr:80,80,80[,rcm]> [launches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from basic_iterative_solvers_amd import Context  # noqa: E402

spec = sys.argv[1]
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = Context()
kind, rest = spec.split(":")
parts = rest.split(",")
rcm = "rcm" in parts
nums = [int(p) for p in parts if p.isdigit()]


def gen():
    if kind == "hpcg":
        A = ctx.gen_hpcg(*nums)
    elif kind == "anderson":
        A = ctx.gen_anderson(nums[0])
    elif kind == "fem":
        A = ctx.gen_fem(*nums)
    else:
        A = ctx.gen_unstr(*nums)
    if rcm:
        B = ctx.permute(A, ctx.bfs_order(A, rcm=True))
        A.free()
        A = B
    return A


ctx.set_option("spmv_valdict", 0)
ref = None
for label, opts in [("rowblock (CRS arrays)", {"spmv_win8": 0}),
                    ("win8 default", {}),
                    ("win8 R=4 D=1", {"spmv_win8_rows": 4, "spmv_win8_depth": 1}),
                    ("win8 R=4 D=2", {"spmv_win8_rows": 4, "spmv_win8_depth": 2}),
                    ("win8 R=4 D=3", {"spmv_win8_rows": 4, "spmv_win8_depth": 3}),
                    ("win8 R=2 D=3", {"spmv_win8_rows": 2, "spmv_win8_depth": 3}),
                    ("win8 R=1 D=3", {"spmv_win8_rows": 1, "spmv_win8_depth": 3})]:
    for k, v in opts.items():
        ctx.set_option(k, v)
    A = gen()
    N = A.n_rows
    x, y = ctx.upload(np.random.default_rng(12345).uniform(-1, 1, N)), ctx.alloc(N)
    for _ in range(3):
        ctx.spmv(A, x, y)
    ctx.sync()
    info = A.spmv_stream_info()
    ctx.profile(True)
    for _ in range(launches):
        ctx.spmv(A, x, y)
    ctx.sync()
    ctx.profile(False)
    n, ms = ctx.profile_read()
    yh = y.to_host()
    if ref is None:
        ref = yh
    alg = 12 * A.nnz + (24 if A.rp_width == 8 else 20) * N
    t = ms / n
    print(f"{spec} {label:24s} form {info[3]} {t:8.4f} ms  algorithmic {alg / t / 1e6:7.1f} GB/s = {alg / t / 8e9:.3f}  "
          f"streamed {A.spmv_streamed_bytes() / 1e9:.3f} GB = {A.spmv_streamed_bytes() / t / 1e6:7.1f} GB/s  identical {np.array_equal(yh, ref)}", flush=True)
    A.free(); x.free(); y.free()
    for k in opts:
        ctx.set_option(k, -1)
ctx.close()
