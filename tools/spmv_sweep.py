#!/usr/bin/env python3
"""Tuning sweep (GPU box): SpMV variants x chunk sizes on HPCG-n, each config
in a fresh subprocess (the variant is latched per process).  Prints GB/s of
algorithmic traffic (12 nnz + 20 N) per config."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
from basic_iterative_solvers_amd import Context
n1 = int(sys.argv[1]); reps = int(sys.argv[2]); kind = sys.argv[3]
ctx = Context(0)
A = ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_anderson(n1)
N = A.n_rows
x, y = ctx.alloc(N), ctx.alloc(N)
ctx.init_vector(x, 1.0)
for _ in range(3): ctx.spmv(A, x, y)
ctx.sync(); ctx.profile(True)
for _ in range(reps): ctx.spmv(A, x, y)
n, ms = ctx.profile_read()
b = 12 * A.nnz + 20 * N
print(json.dumps(dict(ms=ms / n, GBs=b / (ms / n * 1e-3) / 1e9)))
'''


def main():
    n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    kind = sys.argv[2] if len(sys.argv) > 2 else "hpcg"
    variants = [int(v) for v in os.environ.get("SWEEP_VARIANTS", "10,40,41,20,21,140,141").split(",")]
    chunks = [int(c) for c in os.environ.get("SWEEP_CHUNKS", "2048,3072,4096,6144").split(",")]
    for v in variants:
        for c in chunks:
            env = dict(os.environ, BIS_SPMV_VARIANT=str(v), BIS_SPMV_CHUNK=str(c))
            try:
                out = subprocess.run([sys.executable, "-c", CHILD % ROOT, str(n1), "20", kind], env=env,
                                     capture_output=True, text=True, timeout=120)
                res = json.loads(out.stdout.strip().splitlines()[-1])
                print(f"variant {v:4d} chunk {c:5d}: {res['ms']:.4f} ms  {res['GBs']:.0f} GB/s", flush=True)
            except Exception as ex:  # noqa
                print(f"variant {v:4d} chunk {c:5d}: FAILED {ex} {out.stderr[-300:] if 'out' in dir() else ''}", flush=True)


if __name__ == "__main__":
    main()
