#!/usr/bin/env python3
"""bis_mat_tune_placement on HPCG-256: SpMV ms before / after.  python tools/placement_tune_check.py [trials]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import Context
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = Context(0)
for rep in range(3):
    A = ctx.gen_hpcg(256)
    f, b = ctx.tune_placement(A, trials)
    print(f"matrix {rep}: first {f:.4f} ms -> best of {trials} re-allocations {b:.4f} ms", flush=True)
