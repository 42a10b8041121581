#!/usr/bin/env python3
"""Where does the win8 stream have to lie to be read at the fast level?  ONE pool allocation; the stream copied to a series of
offsets inside it and the SpMV timed at each (bis_mat_win8_debug_stream).  python tools/win8_offsets.py [pool GiB] [step MiB]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from basic_iterative_solvers_amd import Context  # noqa: E402

pool_gib = float(sys.argv[1]) if len(sys.argv) > 1 else 24
step_mib = float(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = Context()
ctx.set_option("spmv_valdict", 0)
ctx.set_option("spmv_win8_tune", 0)
n1 = 256
N = n1 ** 3
A = ctx.gen_hpcg(n1)
x, y = ctx.upload(np.random.default_rng(1).uniform(-1, 1, N)), ctx.alloc(N)
ctx.spmv(A, x, y); ctx.sync()
ptr, nbytes = C.c_void_p(), C.c_size_t()
ctx.check(ctx.lib.bis_mat_win8_debug_stream(A.h, C.byref(ptr), C.byref(nbytes), None))
nb = nbytes.value


def t():
    for _ in range(2):
        ctx.spmv(A, x, y)
    ctx.sync(); ctx.profile(True)
    for _ in range(8):
        ctx.spmv(A, x, y)
    ctx.sync(); ctx.profile(False)
    n, ms = ctx.profile_read()
    return ms / n


print(f"library's own buffer at {ptr.value:#x}: {t():.4f} ms ({nb} bytes)", flush=True)
pool = ctx.alloc(int(pool_gib * (1 << 30)) // 8)
print(f"pool at {pool.ptr:#x}, {pool_gib} GiB", flush=True)
step = int(step_mib * (1 << 20))
out = []
off = 0
while off + nb <= pool.n * 8:
    dst = pool.ptr + off
    ctx.check(ctx.lib.bis_copy_vector(ctx.h, C.c_void_p(dst), C.c_void_p(ptr.value), C.c_int64(nb // 8)))
    ctx.check(ctx.lib.bis_mat_win8_debug_stream(A.h, None, None, C.c_void_p(dst)))
    out.append((off, t()))
    off += step
ctx.check(ctx.lib.bis_mat_win8_debug_stream(A.h, None, None, None))
for off, ms in out:
    print(f"offset {off / (1 << 20):9.1f} MiB: {ms:.4f} ms {'FAST' if ms < 0.80 else ''}", flush=True)
print(f"back in the library's buffer: {t():.4f} ms")
ctx.close()
