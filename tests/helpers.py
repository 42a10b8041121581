"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np

from oracle.pyoracle import CRS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_MATS = ["FDM-2d-16", "matrix_band_klein", "hpcg8", "hpcg_4x6x5",
               "anderson8_shift9", "anderson6_raw"]


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"golden_{name}.npz")))


def crs_of(g, prefix):
    rp = g[prefix + "_rp"]
    return CRS(len(rp) - 1, rp, g[prefix + "_col"], g[prefix + "_val"])


def load_histories():
    with open(os.path.join(GOLDEN, "histories.json")) as f:
        return json.load(f)


def load_histories_mid():
    """Mid-size reference histories (tests/golden/make_golden.py --mid-only): inputs are generator strings."""
    with open(os.path.join(GOLDEN, "histories_mid.json")) as f:
        return json.load(f)


def gen_from_cli_arg(orc, arg):
    """The oracle's matrix for a host-CLI generator string (hpcg:N | anderson:L,shift=S | fem:X,Y,Z)."""
    kind, rest = arg.split(":")
    parts = rest.split(",")
    nums = [int(p) for p in parts if "=" not in p]
    kw = {k: float(v) for k, v in (p.split("=") for p in parts if "=" in p)}
    if kind == "hpcg":
        return orc.gen_hpcg(*nums)
    if kind == "anderson":
        return orc.gen_anderson(nums[0], shift=kw.get("shift", 0.0))
    if kind == "unstr":
        return orc.gen_unstr(*nums)
    return orc.gen_fem(*nums)


def parse_hist_key(key):
    name, solver, pc, kws = key.split("|")
    kw = {}
    for item in filter(None, kws.split(",")):
        k, v = item.split("=")
        kw[k] = (v == "True") if v in ("True", "False") else int(v)
    return name, solver, pc, kw


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = np.max(np.abs(b)) if b.size else 1.0
    if denom == 0.0:
        denom = 1.0
    return float(np.max(np.abs(a - b)) / denom) if a.size else 0.0


def hist_dev(h, g):
    """max_k |h_k - g_k| / g_0 over the common window."""
    n = min(len(h), len(g))
    h = np.asarray(h[:n])
    g = np.asarray(g[:n])
    return float(np.max(np.abs(h - g)) / g[0])


# Residual-history parity gate (SURVEY.md section 8d / section 7 "parity
# tolerance").  All deviations are max_k |r_k - r_k^ref| / r_0.
#   CG / Jacobi / GS / SGS      : 1e-10 over the whole history
#   BiCGSTAB                    : 1e-10 over the first 2 iterations, 1e-4 over
#       the whole history -- BiCGSTAB amplifies rounding differences (1e-16
#       -> 6e-7 at iteration 3 on matrix_band_klein -bi -p sgs): the
#       reference itself moves by 9.3e-6 between 1 and 8 threads (HPCG-64,
#       SURVEY.md section 7) and by 6.3e-5 against a differently-rounded
#       restatement on matrix_band_klein (-bi -p gs)
#   GMRES                       : 1e-10 over the whole history (preconditioned
#       residual estimate |g_{j+1}|)
# Iteration counts are compared exactly only where the stopping decision is
# not a rounding tie: TOL = 1e-14 puts the threshold at the rounding floor, so
# two correct implementations may stop one rounding-level iteration apart
# (seen for GMRES on FDM-2d-16: 34 vs 51 iterations with |dr| = 4e-14 r0).
HIST_TOL = {"cg": 1e-10, "j": 1e-10, "gs": 1e-10, "sgs": 1e-10, "gm": 1e-10,
            "bi": 1e-4}


def load_histories_r4():
    """Round-4 reference histories (tests/golden/make_golden.py --r4-only): the first 100 CG iterations on the raw
    (indefinite) Anderson operator, and the unstructured config-5 input with the solver pairs of configs 5 and 4."""
    with open(os.path.join(GOLDEN, "histories_r4.json")) as f:
        return json.load(f)


def permute_crs(A, perm):
    """B = P A P^T for perm[new] = old, entries keeping their order inside a row (what bis_mat_permute builds)."""
    perm = np.asarray(perm, dtype=np.int64)
    n = A.n_rows
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    lens = np.diff(A.row_ptr)[perm]
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    src = np.repeat(A.row_ptr[:-1][perm] - rp[:-1], lens) + np.arange(rp[-1])
    return CRS(n, rp, inv[A.col[src]].astype(np.int32), A.val[src])


def check_history(r, e, solver, scale=1.0, stable_window=False, long_history=False):
    """stable_window: compare only the part of the reference history that is
    independent of rounding (golden `stable_len`: where the reference and the
    oracle -- same algorithm, different rounding -- still agree to 1e-9 r0).
    Used for the GPU path; the few BiCGSTAB runs that sit at a breakdown
    ((r0~, v) cancelling exactly: matrix_band_klein -bi -p gs reaches 38
    iterations in the reference, 30 in the oracle, and an exact 0/0 with a tree
    reduction) cannot be compared past that point by any implementation."""
    g = np.array(e["hist"])
    h = np.asarray(r["hist"])
    if stable_window and e.get("stable_len", len(g)) < len(g):
        n = max(1, min(e["stable_len"], len(h)))
        assert len(h) >= min(e["stable_len"], len(g)) or not np.all(np.isfinite(h))
        assert hist_dev(h[:n], g[:n]) <= 1e-8
        return
    tol = HIST_TOL[solver] * scale
    unstable = bool(np.any(g > 1e3 * g[0])) or not np.all(np.isfinite(g))
    if unstable:
        # divergent / chaotic in the reference itself: compare until the
        # history leaves 1e3 * r0 (at most 50 iterations), loosely
        bad = ~(g <= 1e3 * g[0])
        n = int(np.argmax(bad)) if np.any(bad) else len(g)
        n = max(1, min(n, 50, len(h)))
        assert hist_dev(h[:n], g[:n]) <= 1e-8
        return
    if solver == "bi":
        k = min(3, len(g), len(h))
        assert hist_dev(h[:k], g[:k]) <= 1e-10 * scale
    assert hist_dev(h, g) <= tol
    if e["iters"] is not None and solver in ("cg", "j", "gs", "sgs"):
        n = min(len(h), len(g))
        if len(h) != len(g):
            # TOL = 1e-14 puts the stopping threshold inside the rounding floor, where the iteration at which a run
            # crosses it is a property of its rounding, not of the algorithm: 2 iterations on the small inputs, up to
            # 10 % of a long history.  HPCG-32 -sgs: the reference's last eight residuals wander between 1.63e-12 and
            # 1.84e-12 around the threshold 1.648e-12; it stops at 895, the GPU at 892, 6e-16 r0 apart.  HPCG-48 -cg:
            # the reference (sequential sums of 10^5 terms in its dots) needs 104 iterations, the GPU (tree sums) 96,
            # the two histories 9e-13 r0 apart at most.  The histories themselves are held to `tol` over the common
            # window above, and the shorter run's last residual to `tol` of the longer one's at the same index below.
            # (the wide allowance is for the mid-size inputs only -- `long_history` -- whose stopping iteration is the
            # rounding noise described above; the small goldens hold at 2.  Either way the longer run's extra entries
            # must already sit at the floor the comparison itself resolves: the shorter run has crossed the threshold, the longer one
            # is within `tol` r0 of it at that index, and every residual behind stays below 2 tol r0 + 10 x the threshold.)
            assert abs(len(h) - len(g)) <= (max(2, len(g) // 10) if long_history else 2)
            assert abs(h[n - 1] - g[n - 1]) <= tol * g[0]
            longer = h if len(h) > len(g) else g
            stop = e.get("stopping") or 1e-14 * g[0]
            assert np.all(np.asarray(longer[n - 1:]) <= 2.0 * tol * g[0] + 10.0 * max(stop, 1e-14 * g[0]))
        else:
            assert r["converged"] == e["converged"]
