#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference compiled into
oracle/_ref by `make -C oracle ref`).  Outputs are data only: inputs and the
reference's outputs.  OMP_NUM_THREADS is forced to 1 (the reference's
multi-thread reductions are not run-to-run reproducible, SURVEY.md section 5
defect 4).

  FDM-2d-16.mtx, matrix_band_klein.mtx  -- data files shipped with the
      reference (data/matrices/), copied verbatim as input fixtures
  golden_<name>.npz   -- CRS after the reference's reader, split_LU / peel_diag
      outputs, kernel in/out vectors, ILU(0) factors
  histories.json      -- residual histories of the reference's solvers

Inputs that do not exist as files (HPCG-n, Anderson-L) are produced by the
oracle's generators (oracle/bis_oracle.c: orc_gen_hpcg / orc_gen_anderson) and
fed to the reference as CRS arrays; the generator definitions are this repo's
own, the solver outputs are the reference's.
"""
import json
import os
import sys

os.environ["OMP_NUM_THREADS"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

from oracle.pyoracle import CRS, Oracle, Ref  # noqa: E402


def pack(prefix, M, d):
    d[prefix + "_rp"] = M.row_ptr.astype(np.int64)
    d[prefix + "_col"] = M.col.copy()
    d[prefix + "_val"] = M.val.copy()


def kernel_vectors(ref, A, name):
    d = {}
    n = A.n_rows
    rng = np.random.default_rng(12345)
    x = rng.uniform(-1, 1, n)
    y = rng.uniform(-1, 1, n)
    pack("A", A, d)
    d["x"], d["y"] = x, y
    d["spmv"] = ref.spmv(A, x)
    L, Ls, U, Us = ref.split_LU(A)
    for k, M in zip(("L", "Ls", "U", "Us"), (L, Ls, U, Us)):
        pack(k, M, d)
    D, Dinv = ref.peel_diag(L)
    ref.peel_diag(U)
    pack("Lpeeled", L, d)
    pack("Upeeled", U, d)
    d["A_D"], d["A_D_inv"] = D, Dinv
    d["scale"] = ref.extract_scale(A)
    d["sptrsv"] = ref.sptrsv(Ls, D, y)
    d["bsptrsv"] = ref.sptrsv(Us, D, y, backward=True)
    inpl = y.copy()
    d["sptrsv_inplace"] = ref.sptrsv(Ls, D, inpl, x=inpl)
    d["sub"] = ref.subtract_vectors(x, y, 0.37)
    d["sum"] = ref.sum_vectors(x, y, -1.25)
    d["mul"] = ref.elemwise_mult_vectors(x, y, -1.0)
    d["div"] = ref.elemwise_div_vectors(x, D, 1.0)
    d["dot"] = np.array([ref.dot(x, y)])
    d["norm"] = np.array([ref.norm(x)])
    d["scale_vec"] = ref.scale(x, 1.0 / 3.0)
    d["residual"] = ref.compute_residual(A, x, y)
    d["normalize_x"] = ref.normalize_x(d["spmv"], x, D, y)
    ones = np.ones(n)
    for pc in ("none", "j", "gs", "bgs", "sgs", "2st", "s2st"):
        d["pc_" + pc] = ref.apply_preconditioner(pc, Ls, Us, D, Dinv, ones,
                                                 ones, y)
    d["pc_gs_inplace"] = ref.apply_preconditioner("gs", Ls, Us, D, Dinv, ones,
                                                  ones, y, inplace=True)
    # second build configuration: PRECOND_INNER_ITERS=2, PRECOND_OUTER_ITERS=2
    ref2 = Ref("_i2o2")
    for pc in ("j", "gs", "sgs", "2st", "s2st"):
        d["pc22_" + pc] = ref2.apply_preconditioner(pc, Ls, Us, D, Dinv, ones,
                                                    ones, y)
    # real ILU(0) factors (factor_ILU0_old) and their application
    iLs, iLD, iUs, iUD = ref.factor_ilu0(A)
    pack("iluLs", iLs, d)
    pack("iluUs", iUs, d)
    d["iluLD"], d["iluUD"] = iLD, iUD
    d["pc_ilu0"] = ref.apply_preconditioner("ilu0", iLs, iUs, D, Dinv, iLD,
                                            iUD, y)
    # multi-axpy (dgemm_transpose1 as gmres.hpp:358 uses it), in-bounds n_vec
    V = rng.uniform(-1, 1, (6, n))
    yy = rng.uniform(-1, 1, 6)
    d["V"], d["yy"] = V, yy
    d["multi_axpy5"] = ref.dgemm_transpose1(V, yy, 5)
    np.savez_compressed(os.path.join(HERE, f"golden_{name}.npz"), **d)


RUNS = [  # (solver, precond, extra kwargs)
    ("cg", "none", {}), ("cg", "j", {}), ("cg", "gs", {}), ("cg", "bgs", {}),
    ("cg", "sgs", {}), ("cg", "2st", {}), ("cg", "s2st", {}),
    ("cg", "ilu0", dict(ilu_real=True)), ("cg", "j", dict(num_scale=True)),
    ("bi", "none", {}), ("bi", "j", {}), ("bi", "gs", {}), ("bi", "sgs", {}),
    ("bi", "ilu0", dict(ilu_real=True)),
    ("gm", "none", dict(restart_len=50)), ("gm", "j", dict(restart_len=50)),
    ("gm", "gs", dict(restart_len=50)), ("gm", "sgs", dict(restart_len=50)),
    ("gm", "ilu0", dict(restart_len=50, ilu_real=True)),
    ("gm", "gs", dict(restart_len=10)), ("gm", "none", dict(restart_len=10)),
    ("j", "none", {}), ("gs", "none", {}), ("sgs", "none", {}),
]


def histories(ref, orc, mats):
    out = {}
    for name, A in mats.items():
        for solver, pc, kw in RUNS:
            r = ref.solve(A, solver, pc, **kw)
            key = f"{name}|{solver}|{pc}|" + ",".join(
                f"{k}={v}" for k, v in sorted(kw.items()))
            entry = dict(iters=r["iters"], converged=r["converged"],
                         stopping=r["stopping"],
                         final_true_residual=r["final_true_residual"],
                         hist=[float(v) for v in r["hist"]])
            # How far is this history independent of rounding?  The oracle is the same
            # algorithm with different rounding (fma row sums vs the reference's
            # re-associated SIMD sums): the first index where the two part by more
            # than 1e-9 r0 bounds what any other implementation can be compared on.
            o_all = orc.solve(A, solver, pc, **kw)
            n_all = min(len(o_all["hist"]), len(r["hist"]))
            d_all = np.abs(o_all["hist"][:n_all] - r["hist"][:n_all]) / r["hist"][0]
            bad = np.nonzero(~(d_all <= 1e-9))[0]
            entry["stable_len"] = int(bad[0]) if len(bad) else int(n_all)
            if solver == "gm":
                # GMRES restarts read one word past `y` in the reference
                # (SURVEY.md section 5 defect 1).  Keep a restarted history
                # only if it agrees with the defined semantics (y[m] = 0,
                # oracle) -- i.e. the stray word happened to be 0.0; else keep
                # the part before the first restart.
                o = orc.solve(A, solver, pc, **kw)
                m = kw.get("restart_len", 10)
                n = min(len(o["hist"]), len(r["hist"]))
                dev = np.max(np.abs(o["hist"][:n] - r["hist"][:n])) / r["hist"][0]
                finite = bool(np.all(np.isfinite(r["hist"])))
                same_len = len(o["hist"]) == len(r["hist"])
                clean = dev < 1e-12 and finite and same_len
                entry["restart_clean"] = bool(clean)
                if not clean:
                    # agreeing prefix if the two only stop at different rounding-level
                    # iterations, else the part before the first restart
                    keep = n if dev < 1e-12 else min(m + 1, n)
                    entry["hist"] = entry["hist"][:keep]
                    entry["stable_len"] = min(entry["stable_len"], keep)
                    entry["iters"] = None
                    entry["converged"] = None
                    entry["final_true_residual"] = None
            out[key] = entry
    with open(os.path.join(HERE, "histories.json"), "w") as f:
        json.dump(out, f, indent=0)
    return out


# Mid-size inputs: full residual histories of the real reference at sizes where blocks, windows,
# tiles and levels are many -- the HIP path is run on the same generator inputs (device generators, bit-identical to
# the oracle's: tests/test_gpu_kernels.py) to convergence against them.  histories_mid.json, data only.
MID_RUNS = [
    ("hpcg48", "hpcg:48", lambda o: o.gen_hpcg(48), [("cg", "none", {}), ("cg", "j", {}), ("cg", "sgs", {})]),
    ("hpcg32", "hpcg:32", lambda o: o.gen_hpcg(32), [("gs", "none", {}), ("sgs", "none", {}), ("j", "none", {})]),
    ("anderson32_shift9", "anderson:32,shift=9", lambda o: o.gen_anderson(32, shift=9.0),
     # (restart length 50: no restart before convergence -- a restart reads y[m] out of bounds in the reference,
     # SURVEY.md section 5 defect 1, and the history behind it is not defined)
     [("gm", "gs", dict(restart_len=50)), ("j", "none", {}), ("cg", "sgs", {})]),
    ("fem_12x11x10", "fem:12,11,10", lambda o: o.gen_fem(12, 11, 10),
     [("bi", "ilu0", dict(ilu_real=True)), ("cg", "j", {})]),
]


def histories_mid(ref, orc):
    out = {}
    for name, cli_arg, gen, runs in MID_RUNS:
        A = gen(orc)
        for solver, pc, kw in runs:
            r = ref.solve(A, solver, pc, **kw)
            o = orc.solve(A, solver, pc, **kw)
            n = min(len(o["hist"]), len(r["hist"]))
            d = np.abs(o["hist"][:n] - r["hist"][:n]) / r["hist"][0]
            bad = np.nonzero(~(d <= 1e-9))[0]
            key = f"{name}|{solver}|{pc}|" + ",".join(f"{k}={v}" for k, v in sorted(kw.items()))
            out[key] = dict(cli=cli_arg, rows=A.n_rows, nnz=A.nnz, iters=r["iters"], converged=r["converged"],
                            stopping=r["stopping"], final_true_residual=r["final_true_residual"],
                            hist=[float(v) for v in r["hist"]], stable_len=int(bad[0]) if len(bad) else int(n),
                            oracle_max_dev_over_r0=float(np.max(d)))
            print(key, r["iters"], r["converged"], f"oracle dev {np.max(d):.2e}", flush=True)
    with open(os.path.join(HERE, "histories_mid.json"), "w") as f:
        json.dump(out, f, indent=0)
    return out


# Round 4: (i) raw (indefinite) Anderson, the operator of BASELINE configs 2-3 as named: the first 100 CG iterations
# (SURVEY 8d parity gate item ii; cg.hpp:6-54 call order) -- CG need not converge there, the window is what is compared;
# (ii) the unstructured config-5 input (`unstr:`: no grid, no node blocks) with the solver / preconditioner pairs of configs
# 5 and 4.  histories_r4.json, data only.
R4_RUNS = [
    ("anderson16_raw", "anderson:16", lambda o: o.gen_anderson(16), [("cg", "none", dict(max_iters=100))]),
    ("anderson32_raw", "anderson:32", lambda o: o.gen_anderson(32), [("cg", "none", dict(max_iters=100)),
                                                                      ("cg", "j", dict(max_iters=100))]),
    ("unstr_12x11x10", "unstr:12,11,10", lambda o: o.gen_unstr(12, 11, 10),
     [("bi", "ilu0", dict(ilu_real=True)), ("gm", "gs", dict(restart_len=50)), ("cg", "sgs", {}), ("gs", "none", {}),
      # (added later in round 4: the other solver / preconditioner pairs on the matrix without a grid)
      ("gm", "ilu0", dict(ilu_real=True, restart_len=50)), ("cg", "ilu0", dict(ilu_real=True)), ("bi", "sgs", {}), ("cg", "j", {}),
      ("sgs", "none", {}), ("bi", "2st", {}), ("gm", "s2st", dict(restart_len=50))]),
    ("unstr_20x20x20", "unstr:20,20,20", lambda o: o.gen_unstr(20, 20, 20),
     [("bi", "ilu0", dict(ilu_real=True)), ("gm", "gs", dict(restart_len=50)), ("cg", "sgs", {}), ("gm", "ilu0", dict(ilu_real=True, restart_len=50))]),
]


def histories_r4(ref, orc):
    out = {}
    for name, cli_arg, gen, runs in R4_RUNS:
        A = gen(orc)
        for solver, pc, kw in runs:
            r = ref.solve(A, solver, pc, **kw)
            o = orc.solve(A, solver, pc, **kw)
            n = min(len(o["hist"]), len(r["hist"]))
            d = np.abs(o["hist"][:n] - r["hist"][:n]) / r["hist"][0]
            bad = np.nonzero(~(d <= 1e-9))[0]
            key = f"{name}|{solver}|{pc}|" + ",".join(f"{k}={v}" for k, v in sorted(kw.items()))
            out[key] = dict(cli=cli_arg, rows=A.n_rows, nnz=A.nnz, iters=r["iters"], converged=r["converged"],
                            stopping=r["stopping"], final_true_residual=r["final_true_residual"],
                            hist=[float(v) for v in r["hist"]], stable_len=int(bad[0]) if len(bad) else int(n),
                            oracle_max_dev_over_r0=float(np.max(d)))
            print(key, r["iters"], r["converged"], f"oracle dev {np.max(d):.2e} stable {out[key]['stable_len']}", flush=True)
    with open(os.path.join(HERE, "histories_r4.json"), "w") as f:
        json.dump(out, f, indent=0)
    return out


def main():
    ref, orc = Ref(), Oracle()
    if "--r4-only" in sys.argv:  # leaves the earlier fixtures as they are
        h = histories_r4(ref, orc)
        print(f"wrote {len(h)} round-4 histories")
        return
    if "--mid-only" in sys.argv:  # leaves the small fixtures as they are
        h = histories_mid(ref, orc)
        print(f"wrote {len(h)} mid-size histories")
        return
    mats = {}
    for name in ("FDM-2d-16", "matrix_band_klein"):
        A = ref.read_mtx(os.path.join(HERE, name + ".mtx"))
        mats[name] = A
    mats["hpcg8"] = orc.gen_hpcg(8)
    mats["hpcg_4x6x5"] = orc.gen_hpcg(4, 6, 5)
    mats["anderson8_shift9"] = orc.gen_anderson(8, shift=9.0)
    for name, A in mats.items():
        kernel_vectors(ref, A, name)
    # raw Anderson (indefinite): kernel vectors only + first CG iterations
    kernel_vectors(ref, orc.gen_anderson(6), "anderson6_raw")
    h = histories(ref, orc, mats)
    n_clean = sum(1 for k, v in h.items() if v.get("restart_clean", True))
    print(f"wrote {len(h)} histories ({n_clean} complete) and "
          f"{len(mats) + 1} kernel fixture files")
    histories_mid(ref, orc)
    histories_r4(ref, orc)


if __name__ == "__main__":
    main()
