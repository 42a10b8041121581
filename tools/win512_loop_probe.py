"""Why does the north-star SpMV sometimes run 1.2 ms slower inside the CG loop than in the placement trial on the same stream?
One HPCG-512 matrix (8-byte values streamed), several CG objects on it with the allocations between them shifted by dummy
buffers: per-SpMV time inside each loop against the stand-alone time on scratch vectors."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from basic_iterative_solvers_amd import Context  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = Context()
ctx.set_option("spmv_valdict", 0)
A = ctx.gen_hpcg(n1)
N = A.n_rows
xs, ys = ctx.alloc(N), ctx.alloc(N)
ctx.init_vector(xs, 0.5)


def alone(x, y, reps=6):
    for _ in range(2):
        ctx.spmv(A, x, y)
    ctx.sync(); ctx.profile(True)
    for _ in range(reps):
        ctx.spmv(A, x, y)
    ctx.sync()
    k, ms = ctx.profile_read(); ctx.profile(False)
    return ms / max(k, 1)


print(json.dumps({"form": A.spmv_stream_info(), "tuning": A.win8_tuning(), "alone_scratch_ms": alone(xs, ys)}), flush=True)
dummies = []
for trial in range(5):
    b, x = ctx.alloc(N), ctx.alloc(N)
    ctx.init_vector(b, 1.0); ctx.init_vector(x, 0.1)
    cg = ctx.cg(A, b, x)
    cg.init(0.0)
    cg.iterate(3)
    ctx.sync(); ctx.profile(True)
    cg.iterate(10)
    ctx.sync()
    k, ms = ctx.profile_read(); ctx.profile(False)
    rec = {"trial": trial, "in_loop_spmv_ms": ms / max(k, 1), "launches": k, "alone_on_b_x_ms": alone(b, x), "alone_scratch_ms": alone(xs, ys)}
    print(json.dumps(rec), flush=True)
    cg.free(); b.free(); x.free()
    dummies.append(ctx.alloc((3 + trial) * (1 << 27)))  # 3, 4, 5 ... GiB kept: the next vectors land elsewhere
