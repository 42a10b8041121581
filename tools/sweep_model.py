#!/usr/bin/env python3
"""Design tool (CPU): what the chunked natural-order sweep would cost on a given triangle under a simple timing model
(tools/sweep_model.c), against the level count that bounds the level-scheduled kernels.
    python tools/sweep_model.py fem:80,80,81 | unstr:40,40,40[,rcm] | hpcg:128 | anderson:128  [--backward]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pyoracle import CRS, Oracle  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "tools", "libsweepmodel.so"))
lib.sweep_levels.restype = C.c_int64


def strict_lower(A, backward=False):
    """strict lower triangle in processing order (backward: rows/cols mirrored so that operands have smaller indices)"""
    import scipy.sparse as sp
    M = A.to_scipy().tocsr()
    if backward:
        n = A.n_rows
        p = np.arange(n)[::-1]
        M = M[p][:, p].tocsr()
    T = sp.tril(M, k=-1, format="csr")
    T.sort_indices()
    return T.indptr.astype(np.int64), T.indices.astype(np.int32)


def model(rp, col, chunk_of, W, h, t0, t1, t_chunk=0.0):
    n = len(rp) - 1
    out = np.zeros(4)
    fin = np.zeros(n)
    lib.sweep_model(C.c_int64(n), rp.ctypes, col.ctypes, chunk_of.ctypes, C.c_int(W), C.c_double(h), C.c_double(t0), C.c_double(t1),
                    C.c_double(t_chunk), out.ctypes, fin.ctypes)
    return out


def main():
    spec = sys.argv[1]
    backward = "--backward" in sys.argv
    orc = Oracle()
    kind, rest = spec.split(":")
    parts = rest.split(",")
    nums = [int(p) for p in parts if p.isdigit()]
    t = time.time()
    if kind == "hpcg":
        A = orc.gen_hpcg(*nums)
    elif kind == "anderson":
        A = orc.gen_anderson(nums[0], shift=9.0)
    elif kind == "fem":
        A = orc.gen_fem(*nums)
    else:
        A = orc.gen_unstr(*nums)
    if "rcm" in parts:
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        from helpers import permute_crs
        perm = reverse_cuthill_mckee(A.to_scipy().tocsr(), symmetric_mode=True)
        A = permute_crs(A, perm)
    rp, col = strict_lower(A, backward)
    n = len(rp) - 1
    lvl = np.zeros(n, dtype=np.int32)
    nl = lib.sweep_levels(C.c_int64(n), rp.ctypes, col.ctypes, lvl.ctypes)
    avg = (rp[-1] / n)
    print(f"{spec}{' backward' if backward else ''}: {n} rows, {rp[-1]} entries ({avg:.1f} per row), {nl} levels ({n / nl:.0f} rows per level) "
          f"[{time.time() - t:.1f} s]", flush=True)
    print(f"  level-scheduled at 2.0 us per level: {nl * 2.0 / 1e3:.2f} ms")
    # natural chains: maximal runs of consecutive rows each of which depends on its predecessor, cut at kmax rows
    dep_prev = np.zeros(n, dtype=bool)
    last = col[np.maximum(rp[1:] - 1, 0)]  # ascending columns: the largest operand
    has = rp[1:] > rp[:-1]
    dep_prev[has] = last[has] == (np.arange(n)[has] - 1)
    print(f"  rows that depend on their predecessor: {dep_prev.mean():.3f}")
    for kmax in (8, 32, 128, 1024):
        start = ~dep_prev
        # cut long runs
        run_id = np.cumsum(start) - 1
        first = np.flatnonzero(start)
        pos_in_run = np.arange(n) - first[run_id]
        start |= (pos_in_run % kmax) == 0
        chunk_of = (np.cumsum(start) - 1).astype(np.int32)
        nchunks = int(chunk_of[-1]) + 1
        for (t0, t1, h) in ((0.12, 0.008, 2.0), (0.12, 0.008, 1.2), (0.06, 0.004, 1.2)):
            line = f"  chains kmax={kmax:5d} ({nchunks} chunks, {n / nchunks:.1f} rows each) t={t0}+{t1}len h={h}:"
            for W in (1024, 2048, 4096, 8192):
                o = model(rp, col, chunk_of, W, h, t0, t1, t_chunk=0.3)
                line += f"  W={W}: {o[0] / 1e3:.2f} ms ({int(o[1])}x/{int(o[2])}i)"
            print(line, flush=True)
    if "--fixed" not in sys.argv:
        return
    # rows per step of time t0 + t1 len (us)
    for (t0, t1, h, tag) in ((0.12, 0.008, 2.0, "t_row=0.12+0.008 len, h=2.0"), (0.12, 0.008, 1.2, "h=1.2"), (0.06, 0.004, 1.2, "fast rows, h=1.2")):
        for W in (1024, 4096):
            line = f"  {tag:28s} W={W:5d}:"
            for K in (1, 4, 8, 16, 32, 64, 128, 256):
                chunk_of = (np.arange(n) // K).astype(np.int32)
                o = model(rp, col, chunk_of, W, h, t0, t1, t_chunk=0.3)
                line += f"  K={K}: {o[0] / 1e3:.2f} ms ({int(o[1])}x/{int(o[2])}i)"
            print(line, flush=True)


if __name__ == "__main__":
    main()
