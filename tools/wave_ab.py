"""A/B of the level-scheduled sweep's variants on `unstr:80,80,80` as generated (135 wide levels, no locality)."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from basic_iterative_solvers_amd import Context  # noqa: E402

ctx = Context()
A = ctx.gen_unstr(80, 80, 80)
N = A.n_rows
Ls, Us, D, Dinv = ctx.split_strict(A)
b = ctx.upload(np.random.default_rng(21).uniform(-1, 1, N))
ref = None
for name, opts in (("default", {}), ("wave per row, 8 workgroups per CU", {"trsv_wave": 1, "trsv_wave_wgs": 8}), ("wave per row, 6 per CU", {"trsv_wave": 1, "trsv_wave_wgs": 6}), ("wave per row, 4 per CU", {"trsv_wave": 1}), ("wave per row, 2 per CU", {"trsv_wave": 1, "trsv_wave_wgs": 2}), ("lane per row (trsv_wave 0)", {"trsv_wave": 0}), ("wave, batch 16", {"trsv_batch": 16}), ("wave, batch 4", {"trsv_batch": 4}),
                   ("scratch in row order (trsv_by_pos 0)", {"trsv_by_pos": 0}), ("a launch per level (trsv_grid 0?)", {"trsv_one_xcd": 2})):
    for k, v in opts.items():
        ctx.set_option(k, v)
    try:
        L2, U2, D2, Di2 = ctx.split_strict(A)
        x = ctx.alloc(N)
        out = {}
        for T, solve, d in ((L2, ctx.sptrsv, "forward"), (U2, ctx.bsptrsv, "backward")):
            solve(T, x, D2, b)
            ctx.sync(); ctx.profile(True)
            for _ in range(8):
                solve(T, x, D2, b)
            ctx.sync()
            n, ms = ctx.profile_read_sweeps(); ctx.profile(False)
            out[d] = round(ms / max(n, 1), 4)
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
    except Exception as ex:  # noqa: BLE001
        print(json.dumps({"variant": name, "error": repr(ex)[:200]}), flush=True)
    for k in opts:
        ctx.set_option(k, -1)
