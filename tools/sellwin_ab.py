#!/usr/bin/env python3
"""The compressed SpMV forms of one stencil matrix side by side, SpMV alone and inside the fused CG iteration; y compared bit for bit.
   python tools/sellwin_ab.py hpcg 256 "masks=0" "masks=-1" "masks=-1,rows=1" """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import Context, load_library
import numpy as np

kind, n1 = sys.argv[1], int(sys.argv[2])
cfgs = [dict(kv.split("=") for kv in c.split(",")) for c in sys.argv[3:]]
lib = load_library()
ctx = Context(0)
ref = None
for c in cfgs:
    for k in ("masks", "pairs", "joint", "rows", "nt"):
        lib.bis_set_option(("spmv_sellwin_" + k).encode(), int(c.get(k, -1)))
    A = ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_anderson(n1, shift=9.0)
    N = A.n_rows
    x, y, b = ctx.alloc(N), ctx.alloc(N), ctx.alloc(N)
    x.set(np.random.default_rng(0).uniform(-1, 1, N))
    for _ in range(3): ctx.spmv(A, x, y)
    info = A.spmv_stream_info()
    ctx.sync(); ctx.profile(True)
    for _ in range(20): ctx.spmv(A, x, y)
    n, ms = ctx.profile_read(); ctx.profile(False)
    yh = y.to_host()
    same = ref is None or np.array_equal(ref.view(np.uint64), yh.view(np.uint64))
    if ref is None: ref = yh
    ctx.init_vector(b, 1.0); ctx.init_vector(x, 0.1)
    cg = ctx.cg(A, b, x); cg.init(0.0); cg.iterate(10); ctx.sync()
    t0 = time.perf_counter(); cg.iterate(100); ctx.sync(); dt = time.perf_counter() - t0
    print(f"{c}: stream info {info}, streamed {A.spmv_streamed_bytes() / 1e6:.1f} MB, SpMV {ms / n:.4f} ms, CG {100 / dt:.0f} it/s, y identical to the first: {same}", flush=True)
    cg.free(); A.free(); x.free(); y.free(); b.free()
