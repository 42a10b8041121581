"""The sweep kernels on `unstr:80,80,80` RCM-ordered (7119 narrow levels): chained (default), wave per row, lane per row."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from basic_iterative_solvers_amd import Context  # noqa: E402

ctx = Context()
A0 = ctx.gen_unstr(80, 80, 80)
A = ctx.permute(A0, ctx.bfs_order(A0, rcm=True))
A0.free()
N = A.n_rows
b = ctx.upload(np.random.default_rng(21).uniform(-1, 1, N))
ref = None
for name, opts in (("default (chained)", {}), ("level-scheduled, wave per row", {"trsv_chain": 0, "trsv_wave": 1}),
                   ("level-scheduled, wave per row, 6 per CU", {"trsv_chain": 0, "trsv_wave": 1, "trsv_wave_wgs": 6}), ("level-scheduled, wave per row, 2 per CU", {"trsv_chain": 0, "trsv_wave": 1, "trsv_wave_wgs": 2}), ("level-scheduled, wave per row, 1 per CU", {"trsv_chain": 0, "trsv_wave": 1, "trsv_wave_wgs": 1}), ("level-scheduled, trial", {"trsv_chain": 0})):
    for k, v in opts.items():
        ctx.set_option(k, v)
    L2, U2, D2, Di2 = ctx.split_strict(A)
    x = ctx.alloc(N)
    out = {}
    for T, solve, d in ((L2, ctx.sptrsv, "forward"), (U2, ctx.bsptrsv, "backward")):
        solve(T, x, D2, b)
        ctx.sync(); ctx.profile(True)
        for _ in range(4):
            solve(T, x, D2, b)
        ctx.sync()
        n, ms = ctx.profile_read_sweeps(); ctx.profile(False)
        out[d] = round(ms / max(n, 1), 3)
        out[d + "_kernel"] = T.sweep_kernel(d == "backward")
        if d == "forward":
            xh = x.to_host()
            if ref is None:
                ref = xh
            out["identical"] = bool(np.array_equal(ref.view(np.uint64), xh.view(np.uint64)))
    print(json.dumps({"variant": name, **out}), flush=True)
    for m in (L2, U2):
        m.free()
    x.free(); D2.free(); Di2.free()
    for k in opts:
        ctx.set_option(k, -1)
