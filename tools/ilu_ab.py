"""Persistent ILU(0): workgroups per CU, on unstr:80,80,80 as generated and RCM-ordered, and the FEM-like matrix."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from basic_iterative_solvers_amd import Context  # noqa: E402

ctx = Context()
for kind in ("asis", "rcm", "fem"):
    if kind == "fem":
        A = ctx.gen_fem(80, 80, 81)
    else:
        A = ctx.gen_unstr(80, 80, 80)
        if kind == "rcm":
            B = ctx.permute(A, ctx.bfs_order(A, rcm=True)); A.free(); A = B
    ref = None
    for wgs in (4, 2, 6, 7):
        ctx.set_option("ilu0_wgs", wgs)
        best = 1e9
        for rep in range(2):
            ctx.sync(); t0 = time.perf_counter()
            Ls, Us, LD, UD = ctx.ilu0(A)
            ctx.sync(); best = min(best, time.perf_counter() - t0)
            if rep == 0:
                v = Ls.download()[2]
                if ref is None:
                    ref = v
                same = bool(np.array_equal(ref.view(np.uint64), v.view(np.uint64)))
            Ls.free(); Us.free(); LD.free(); UD.free()
        print(json.dumps({"matrix": kind, "ilu0_wgs": wgs, "whole_call_ms": round(best * 1e3, 2), "identical_L": same}), flush=True)
    ctx.set_option("ilu0_wgs", -1)
    A.free()
