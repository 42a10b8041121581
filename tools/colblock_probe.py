"""Experiment: does a matrix WITHOUT locality (unstr:80,80,80 as generated) multiply faster when its columns are taken
slab by slab (K passes, each pass gathering from an x slice that fits the L2 of an XCD)?  Times the existing SpMV kernels on
the K column slabs of the matrix (the sum of the passes is what a chained-accumulator kernel would cost at best) against
the one-pass SpMV.  Not a product path."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from basic_iterative_solvers_amd import Context  # noqa: E402
from oracle.pyoracle import CRS  # noqa: E402


def timed(ctx, A, x, y, reps=20):
    for _ in range(3):
        ctx.spmv(A, x, y)
    ctx.sync()
    ctx.profile(True)
    for _ in range(reps):
        ctx.spmv(A, x, y)
    ctx.sync()
    k, ms = ctx.profile_read()
    ctx.profile(False)
    return ms / max(k, 1)


def main():
    ctx = Context()
    A = ctx.gen_unstr(80, 80, 80)
    n = A.n_rows
    x, y = ctx.alloc(n), ctx.alloc(n)
    x.set(np.random.default_rng(1).uniform(-1, 1, n))
    out = {"n": n, "nnz": A.nnz, "one_pass_ms": timed(ctx, A, x, y), "one_pass_form": A.spmv_stream_info()}
    print(json.dumps(out), flush=True)
    rp, col, val = A.download()
    rp = rp.astype(np.int64)
    row_of = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    for K in (2, 3, 4, 6, 8, 12):
        edges = np.linspace(0, n, K + 1).astype(np.int64)
        slab = np.searchsorted(edges, col, side="right") - 1
        total, parts = 0.0, []
        for k in range(K):
            m = slab == k
            cnt = np.bincount(row_of[m], minlength=n)
            rpk = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
            B = ctx.matrix(CRS(n, rpk, col[m].astype(np.int32), val[m]))
            t = timed(ctx, B, x, y)
            parts.append(round(t, 4))
            total += t
            B.free()
        rec = {"K": K, "slab_MB": round(8 * n / K / 1e6, 2), "sum_ms": round(total, 4), "parts_ms": parts}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
