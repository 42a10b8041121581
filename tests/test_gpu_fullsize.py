"""GPU: DIRECT oracle parity at the BASELINE sizes (round-4 verdict, items 1 / 3 / 4).

The full-size gates of tests/test_gpu_kernels.py are size-independent properties (closed forms, symmetry, a triangular
solve undone by a product); a symmetric error passes them.  The oracle's generators produce any row range [r0, r1) of the
same matrices bit-identically (tests/test_gpu_kernels.py::test_*_generator_bit_exact with row0/row1), so here the device's
y = A x at HPCG-256 / HPCG-512 / Anderson-256 (raw and shift 9) / fem:80,80,81 / unstr:80,80,80 is compared, slab by slab
of 10^5 rows, with orc_spmv (reference kernels.hpp:22-42) on the oracle's own rows -- for both stream formats (the CRS
value stream of the headline, and the library's default format for the matrix).

Config 5 AS NAMED (unstructured, 1,536,000 rows, -bi -p ilu0) at its size: generator bit-exact, the RCM ordering, the
chained sweeps (bis_trsv_chain.hip: the residency / straddle bound is size-dependent) bit-exact against the serial oracle on
the whole triangle, device ILU(0) by its defining property on sampled rows, its apply bit-exact, and the solve through the
host CLI with the TRUE residual of the returned x* recomputed by the oracle.
Reference: kernels.hpp:22-42, :54-117, :386-394; methods/bicgstab.hpp:8-83."""
import os
import subprocess
import re

import numpy as np
import pytest

from helpers import permute_crs
from oracle.pyoracle import CRS

pytestmark = pytest.mark.gpu

KTOL = 1e-13
SLAB = 100_000
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "basic_iterative_solvers_amd", "host", "basic_iterative_solvers")


@pytest.fixture(scope="module")
def ctx():
    from basic_iterative_solvers_amd import Context
    c = Context()
    yield c
    c.close()


def slabs_of(N, seed):
    """first, last and three seeded slabs of SLAB rows (unaligned starts)"""
    rng = np.random.default_rng(seed)
    starts = [0, N - SLAB] + [int(s) for s in rng.integers(1, N - SLAB - 1, 3)]
    return [(s, s + SLAB) for s in starts]


def crs_rows(A, r0, r1):
    """rows [r0, r1) of a host CRS as a slab with global columns"""
    s, e = int(A.row_ptr[r0]), int(A.row_ptr[r1])
    return CRS(r1 - r0, (A.row_ptr[r0:r1 + 1] - s).astype(np.int64), A.col[s:e], A.val[s:e], n_cols=A.n_cols)


def check_slabs(ctx, oracle, dA, slab_of, seed, label):
    """y = A x on the device under both stream formats against orc_spmv on the oracle's rows of each slab"""
    N = dA.n_rows
    x = np.random.default_rng(seed).uniform(-1, 1, N)
    dx, dy = ctx.upload(x), ctx.alloc(N)
    refs = []
    for (r0, r1) in slabs_of(N, seed):
        S = slab_of(r0, r1)
        assert S.n_rows == r1 - r0
        refs.append((r0, r1, oracle.spmv(S, x)))
    seen = []
    try:
        for valdict in (0, -1):  # 0: the 8-byte CRS values streamed (the headline's kernel); -1: the library's default format
            ctx.set_option("spmv_valdict", valdict)
            ctx.init_vector(dy, float("nan"))
            ctx.spmv(dA, dx, dy)
            ctx.sync()
            col_b, val_b, n_dict, form = dA.spmv_stream_info()
            seen.append((valdict, form, val_b))
            for r0, r1, yr in refs:
                y = dy.offset(r0, r1 - r0).to_host()
                err = np.max(np.abs(y - yr)) / np.max(np.abs(yr))
                assert err <= KTOL, f"{label} rows [{r0},{r1}) format {form} ({val_b}-byte values): {err:.2e}"
    finally:
        ctx.set_option("spmv_valdict", -1)
    assert seen[0][2] == 8  # the first pass did stream the CRS values
    dx.free(); dy.free()
    return seen


CASES = {
    "hpcg256": (lambda c: c.gen_hpcg(256), lambda o, a, b: o.gen_hpcg(256, row0=a, row1=b)),
    "hpcg512": (lambda c: c.gen_hpcg(512), lambda o, a, b: o.gen_hpcg(512, row0=a, row1=b)),
    "anderson256_raw": (lambda c: c.gen_anderson(256), lambda o, a, b: o.gen_anderson(256, row0=a, row1=b)),
    "anderson256_shift9": (lambda c: c.gen_anderson(256, shift=9.0), lambda o, a, b: o.gen_anderson(256, shift=9.0, row0=a, row1=b)),
    "fem80x80x81": (lambda c: c.gen_fem(80, 80, 81), lambda o, a, b: o.gen_fem(80, 80, 81, row0=a, row1=b)),
}


@pytest.mark.parametrize("case", list(CASES))
def test_sampled_slab_oracle_parity(ctx, oracle, case):
    """A[r0:r1] x on the device == orc_spmv(oracle's rows r0..r1, x) to 1e-13, five slabs of 10^5 rows, both stream formats."""
    dev_gen, orc_gen = CASES[case]
    dA = dev_gen(ctx)
    seen = check_slabs(ctx, oracle, dA, lambda a, b: orc_gen(oracle, a, b), seed=77, label=case)
    if case.startswith("hpcg") or case.startswith("anderson"):
        assert seen[1][2] < 8, f"{case}: the default format should be a compressed one, got {seen}"
    dA.free()


@pytest.fixture(scope="module")
def unstr_full(oracle):
    return oracle.gen_unstr(80, 80, 80)


def test_config5_unstructured_full_size_generator_and_spmv(ctx, oracle, unstr_full):
    """unstr:80,80,80 (1,536,000 rows, ~1.0e8 non-zeros, no grid hint): the device generator equals the oracle's bit for
    bit at this size, and y = A x (an x without any locality) equals orc_spmv on sampled slabs."""
    A = unstr_full
    dA = ctx.gen_unstr(80, 80, 80)
    assert dA.n_rows == 1_536_000 == A.n_rows
    rp, col, val = dA.download()
    assert np.array_equal(rp, A.row_ptr) and np.array_equal(col, A.col) and np.array_equal(val, A.val)
    del rp, col, val
    check_slabs(ctx, oracle, dA, lambda a, b: crs_rows(A, a, b), seed=78, label="unstr80")
    dA.free()


def _product_check(ctx, T, D, x, b_dev, N, tol):
    t, u = ctx.alloc(N), ctx.alloc(N)
    ctx.spmv(T, x, t)
    ctx.elemwise_mult_vectors(u, D, x)
    ctx.sum_vectors(t, t, u)
    ctx.subtract_vectors(t, t, b_dev)
    r = ctx.euclidean_vec_norm(t)
    t.free(); u.free()
    assert r <= tol, r


def test_config5_unstructured_full_size_rcm_sweeps_ilu0_and_solve(ctx, oracle, unstr_full, tmp_path, capfd, monkeypatch):
    """Config 5 as named, RCM-ordered (the banded order a real mesh is solved in: ~7 thousand dependency levels):
    (1) P A P^T on the device == the oracle's matrix permuted on the host, bit for bit; slab SpMV parity on it;
    (2) forward / backward sweeps of the strict triangles -- the chained sweep at this size -- bit-exact against the
        serial oracle (kernels.hpp:54-117) on the WHOLE triangle, x aliasing b included, and (D + T) x == b;
    (3) device ILU(0): (L U)_ij == A_ij on A's pattern for sampled rows; its apply (kernels.hpp:386-394) bit-exact against
        the oracle on the device's factors;
    (4) `-bi -p ilu0 -perm rcm` through the host CLI converges, and the TRUE residual of the x* it returns (caller's order),
        recomputed by the oracle on the oracle's matrix, is <= 1e-9 r0."""
    import scipy.sparse as sp
    import time
    t_start = time.perf_counter()

    def stamp(what):  # (shown with `pytest -s`: where a 1.5 M-row test spends its time)
        print(f"[config5 full size] {what}: {time.perf_counter() - t_start:.2f} s", flush=True)
    A = unstr_full
    N = A.n_rows
    dA0 = ctx.gen_unstr(80, 80, 80)
    perm = ctx.bfs_order(dA0, rcm=True)
    assert np.array_equal(np.sort(perm), np.arange(N))
    dA = ctx.permute(dA0, perm)
    dA0.free()
    B = permute_crs(A, perm)
    rp, col, val = dA.download()
    assert np.array_equal(rp, B.row_ptr) and np.array_equal(col, B.col) and np.array_equal(val, B.val)
    del rp, col, val
    stamp(f"RCM order + P A P^T bit-exact ({B.nnz} non-zeros)")
    check_slabs(ctx, oracle, dA, lambda a, b: crs_rows(B, a, b), seed=79, label="unstr80-rcm")
    stamp("slab SpMV parity")

    # (2) the sweeps
    monkeypatch.setenv("BIS_TRSV_CHAIN_STATS", "1")
    dLs, dUs, dD, dDinv = ctx.split_strict(dA)
    L, Ls, U, Us = oracle.split_LU(B)
    D, _, _ = oracle.peel_diag(L)
    del L, U
    b = np.random.default_rng(21).uniform(-1, 1, N)
    db, x = ctx.upload(b), ctx.alloc(N)
    for solve, dT, T, backward in ((ctx.sptrsv, dLs, Ls, False), (ctx.bsptrsv, dUs, Us, True)):
        want = oracle.sptrsv(T, D, b, backward=backward)
        solve(dT, x, dD, db)
        ctx.sync()
        assert np.array_equal(x.to_host(), want)
        _product_check(ctx, dT, dD, x, db, N, 1e-12 * np.sqrt(N) * 20)
        ctx.copy_vector(x, db)
        solve(dT, x, dD, x)  # x aliases b (gmres.hpp:173)
        assert np.array_equal(x.to_host(), want)
    stamp("sweeps forward / backward / aliased, bit-exact")
    err = capfd.readouterr().err
    print(err, flush=True)
    assert err.count(": used") >= 2, "the chained sweep's plan should apply to both triangles at this size:\n" + err[-1500:]
    for m in (dLs, dUs, dDinv):
        m.free()
    del Ls, Us

    # (3) ILU(0)
    fLs, fL_D, fUs, fU_D = ctx.ilu0(dA)
    lrp, lcol, lval = fLs.download()
    urp, ucol, uval = fUs.download()
    uD = fU_D.to_host()
    rng = np.random.default_rng(6)
    rows = np.sort(rng.choice(N, 400, replace=False))
    As = sp.csr_matrix((B.val, B.col, B.row_ptr), shape=(N, N))
    Lm = sp.csr_matrix((lval, lcol, lrp), shape=(N, N)) + sp.identity(N, format="csr")
    Um = sp.csr_matrix((uval, ucol, urp), shape=(N, N)) + sp.diags(uD, format="csr")
    P = (Lm[rows] @ Um).tocsr()
    Ar = As[rows].tocsr()
    worst = 0.0
    for i in range(len(rows)):
        cols = Ar.indices[Ar.indptr[i]:Ar.indptr[i + 1]]
        want = Ar.data[Ar.indptr[i]:Ar.indptr[i + 1]]
        got = np.asarray(P[i, cols].todense()).ravel()
        worst = max(worst, np.abs(got - want).max())
    assert worst <= 1e-12, worst
    del As, Lm, Um, P, Ar
    stamp(f"ILU(0) + (LU)_ij == A_ij on 400 rows (worst {worst:.1e})")
    # its apply: z = U^-1 L^-1 r through the device sweeps, against the serial oracle on the same factors
    hLs, hUs = CRS(N, lrp, lcol, lval), CRS(N, urp, ucol, uval)
    t_ref = oracle.sptrsv(hLs, np.ones(N), b)
    z_ref = oracle.sptrsv(hUs, uD, t_ref, backward=True)
    t = ctx.alloc(N)
    ctx.sptrsv(fLs, t, fL_D, db)
    ctx.bsptrsv(fUs, x, fU_D, t)
    assert np.array_equal(t.to_host(), t_ref)
    assert np.array_equal(x.to_host(), z_ref)
    for m in (fLs, fUs, dA):
        m.free()

    stamp("ILU(0) apply bit-exact")
    # (4) the solve, through the host CLI (the library's own BiCGSTAB schedule, bicgstab.hpp:8-83)
    fx = str(tmp_path / "x.txt")
    out = subprocess.run([BIN, "unstr:80,80,80", "-bi", "-p", "ilu0", "-perm", "rcm", "-dump-x", fx],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"(converged in: |did not converge after )(\d+) iterations", out.stdout)
    assert m and m.group(1).startswith("converged"), out.stdout[-1500:]
    assert int(m.group(2)) <= 60
    xs = np.loadtxt(fx)
    assert xs.shape == (N,)
    ones = np.ones(N)
    r0 = np.linalg.norm(ones - oracle.spmv(A, np.full(N, 0.1)))
    r = np.linalg.norm(ones - oracle.spmv(A, xs))
    stamp(f"CLI solve: {m.group(2)} iterations, true residual {r:.3e} = {r / r0:.2e} r0")
    assert r <= 1e-9 * r0, (r, r0)
