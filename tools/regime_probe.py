#!/usr/bin/env python3
"""One process: stand-alone SpMV back to back; with a 1 GB write between launches; with a 1 GB read between
launches; inside the CG loop.  Tells whether the fast regime is cache warmth carried between launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from basic_iterative_solvers_amd import Context
ctx = Context(0)
A = ctx.gen_hpcg(256)
N = A.n_rows
x, y = ctx.alloc(N), ctx.alloc(N)
x.set(np.random.default_rng(0).uniform(-1, 1, N))
big = ctx.alloc(128 * 1024 * 1024)  # 1 GiB
def series(label, between):
    out = []
    for c in range(8):
        tot = 0.0
        for _ in range(10):
            between()
            ctx.profile(True)
            ctx.spmv(A, x, y)
            n, ms = ctx.profile_read(); ctx.profile(False)
            tot += ms
        out.append(tot / 10)
    print(f"{label:34s}", " ".join(f"{v:.3f}" for v in out), flush=True)
for _ in range(40): ctx.spmv(A, x, y)
series("back to back", lambda: None)
series("1 GiB written between launches", lambda: ctx.init_vector(big, 1.0))
series("1 GiB read between launches", lambda: ctx.euclidean_vec_norm(big))
series("x rewritten between launches", lambda: ctx.scale(x, x, 1.0) if hasattr(ctx, "scale") else ctx.copy_vector(y, x))
series("back to back again", lambda: None)
b, xx = ctx.alloc(N), ctx.alloc(N)
ctx.init_vector(b, 1.0); ctx.init_vector(xx, 0.1)
cg = ctx.cg(A, b, xx, None); cg.init(0.0); cg.iterate(20)
out = []
for c in range(8):
    ctx.profile(True); cg.iterate(10); n, ms = ctx.profile_read(); ctx.profile(False)
    out.append(ms / n)
print(f"{'inside the CG loop (fused dot)':34s}", " ".join(f"{v:.3f}" for v in out), flush=True)
