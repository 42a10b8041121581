"""GPU parity tests proper: the HIP path, called through the C ABI, against
the oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest

from helpers import (GOLDEN, GOLDEN_MATS, check_history, crs_of, load_golden,
                     load_histories, parse_hist_key, relerr)
from oracle.pyoracle import CRS

pytestmark = pytest.mark.gpu

KTOL = 1e-13  # kernel-level relative tolerance, SURVEY.md 8d


@pytest.fixture(scope="module")
def ctx():
    from basic_iterative_solvers_amd import Context
    c = Context()
    info = c.device_info()
    assert info["arch"].startswith("gfx950")
    yield c
    c.close()


def dev_spmv(ctx, A, x):
    dA = ctx.matrix(A)
    dx, dy = ctx.upload(x), ctx.alloc(A.n_rows)
    ctx.spmv(dA, dx, dy)
    y = dy.to_host()
    dA.free(); dx.free(); dy.free()
    return y


@pytest.mark.parametrize("name", GOLDEN_MATS)
def test_kernels_vs_reference_golden(ctx, name):
    g = load_golden(name)
    A, Ls, Us = crs_of(g, "A"), crs_of(g, "Ls"), crs_of(g, "Us")
    n = A.n_rows
    x, y, D, Dinv = g["x"], g["y"], g["A_D"], g["A_D_inv"]
    dA, dLs, dUs = ctx.matrix(A), ctx.matrix(Ls), ctx.matrix(Us)
    dx, dy, dD, dDinv = ctx.upload(x), ctx.upload(y), ctx.upload(D), ctx.upload(Dinv)
    out, tmp, work = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    ones = ctx.upload(np.ones(n))

    ctx.spmv(dA, dx, out)
    assert relerr(out.to_host(), g["spmv"]) <= KTOL
    ctx.sptrsv(dLs, out, dD, dy)
    assert relerr(out.to_host(), g["sptrsv"]) <= KTOL
    ctx.bsptrsv(dUs, out, dD, dy)
    assert relerr(out.to_host(), g["bsptrsv"]) <= KTOL
    ctx.copy_vector(out, dy)
    ctx.sptrsv(dLs, out, dD, out)  # x aliases b (gmres.hpp:173)
    assert relerr(out.to_host(), g["sptrsv_inplace"]) <= KTOL

    # elementwise: bit-exact with the compiled reference
    ctx.subtract_vectors(out, dx, dy, 0.37)
    assert np.array_equal(out.to_host(), g["sub"])
    ctx.sum_vectors(out, dx, dy, -1.25)
    assert np.array_equal(out.to_host(), g["sum"])
    ctx.elemwise_mult_vectors(out, dx, dy, -1.0)
    assert np.array_equal(out.to_host(), g["mul"])
    ctx.elemwise_div_vectors(out, dx, dD, 1.0)
    assert np.array_equal(out.to_host(), g["div"])
    ctx.scale(out, dx, 1.0 / 3.0)
    assert np.array_equal(out.to_host(), g["scale_vec"])
    ctx.copy_vector(out, dx)
    assert np.array_equal(out.to_host(), x)
    ctx.init_vector(out, 0.1)
    assert np.array_equal(out.to_host(), np.full(n, 0.1))
    # aliasing: result == operand (gauss_seidel.hpp:34, kernels.hpp:325,369)
    ctx.copy_vector(out, dy)
    ctx.subtract_vectors(out, dx, out, 0.37)
    assert np.array_equal(out.to_host(), g["sub"])
    ctx.copy_vector(out, dx)
    ctx.elemwise_mult_vectors(out, out, dy, -1.0)
    assert np.array_equal(out.to_host(), g["mul"])

    assert abs(ctx.dot(dx, dy) - g["dot"][0]) <= KTOL * np.abs(x).dot(np.abs(y))
    assert abs(ctx.euclidean_vec_norm(dx) - g["norm"][0]) <= KTOL * g["norm"][0]
    ctx.compute_residual(dA, dx, dy, out, tmp)
    assert relerr(out.to_host(), g["residual"]) <= KTOL
    xn = ctx.upload(g["spmv"])
    ctx.normalize_x(xn, dx, dD, dy)
    assert relerr(xn.to_host(), g["normalize_x"]) <= KTOL

    for pc in ("none", "j", "gs", "bgs", "sgs", "2st", "s2st"):
        inp = ctx.upload(y)
        ctx.apply_preconditioner(pc, n, dLs, dUs, dD, dDinv, ones, ones, out, inp, tmp, work)
        assert relerr(out.to_host(), g["pc_" + pc]) <= KTOL, pc
        assert np.array_equal(inp.to_host(), y)  # input untouched
        inp.free()
    inp = ctx.upload(y)
    ctx.apply_preconditioner("gs", n, dLs, dUs, dD, dDinv, ones, ones, inp, inp, tmp, work)
    assert relerr(inp.to_host(), g["pc_gs_inplace"]) <= KTOL
    for pc in ("j", "gs", "sgs", "2st", "s2st"):  # PRECOND_OUTER_ITERS=2, INNER=2
        inp.set(y)
        ctx.apply_preconditioner(pc, n, dLs, dUs, dD, dDinv, ones, ones, out, inp, tmp, work,
                                 outer=2, inner=2)
        assert relerr(out.to_host(), g["pc22_" + pc]) <= 1e-12, pc
        assert np.array_equal(inp.to_host(), y)
    # ILU(0) apply with the reference's own factors
    iLs, iUs = ctx.matrix(crs_of(g, "iluLs")), ctx.matrix(crs_of(g, "iluUs"))
    iLD, iUD = ctx.upload(g["iluLD"]), ctx.upload(g["iluUD"])
    inp.set(y)
    ctx.apply_preconditioner("ilu0", n, iLs, iUs, dD, dDinv, iLD, iUD, out, inp, tmp, work)
    assert relerr(out.to_host(), g["pc_ilu0"]) <= 1e-12

    V = ctx.upload(g["V"].ravel())
    ctx.multi_axpy(V, n, g["yy"], 5, out, n)
    assert relerr(out.to_host(), g["multi_axpy5"]) <= KTOL


@pytest.mark.parametrize("shape", [(8, 8, 8), (4, 6, 5), (1, 1, 1), (2, 3, 1), (17, 9, 11)])
def test_hpcg_generator_bit_exact(ctx, oracle, shape):
    nx, ny, nz = shape
    N = nx * ny * nz
    for row0, row1 in [(0, N), (N // 3, N - N // 4), (0, 0)]:
        ref = oracle.gen_hpcg(nx, ny, nz, row0, row1)
        d = ctx.gen_hpcg(nx, ny, nz, row0, row1)
        rp, col, val = d.download()
        assert d.n_rows == row1 - row0 and d.n_cols == N
        assert np.array_equal(rp, ref.row_ptr)
        assert np.array_equal(col, ref.col)
        assert np.array_equal(val, ref.val)
        d.free()


@pytest.mark.parametrize("L,shift", [(3, 0.0), (8, 9.0), (13, 0.0)])
def test_anderson_generator_bit_exact(ctx, oracle, L, shift):
    N = L ** 3
    for row0, row1 in [(0, N), (N // 5, N // 2)]:
        ref = oracle.gen_anderson(L, shift=shift, row0=row0, row1=row1)
        d = ctx.gen_anderson(L, shift=shift, row0=row0, row1=row1)
        rp, col, val = d.download()
        assert np.array_equal(rp, ref.row_ptr)
        assert np.array_equal(col, ref.col)
        assert np.array_equal(val, ref.val)
        d.free()


@pytest.mark.parametrize("shape,keep,rows", [((6, 5, 4), 85, None), ((3, 3, 3), 100, None), ((5, 4, 7), 60, (37, 301)),
                                             ((1, 1, 1), 85, None), ((9, 1, 2), 0, None)])
def test_fem_generator_bit_exact(ctx, oracle, shape, keep, rows):
    """Config-5 stand-in (FEM-like, 3 unknowns per node, ragged rows): the device
    generator reproduces the oracle's CRS bit for bit, also for a row range."""
    N = 3 * shape[0] * shape[1] * shape[2]
    row0, row1 = rows if rows else (0, N)
    ref = oracle.gen_fem(*shape, keep=keep, seed=7, row0=row0, row1=row1)
    d = ctx.gen_fem(*shape, keep=keep, seed=7, row0=row0, row1=row1)
    rp, col, val = d.download()
    assert np.array_equal(rp, ref.row_ptr) and np.array_equal(col, ref.col)
    assert np.array_equal(val, ref.val)
    d.free()


def test_fem_medium_kernels_and_ilu0_vs_oracle(ctx, oracle):
    """SpMV, both triangular solves and the device ILU(0) on the ragged FEM-like
    matrix (rows of 18..81 non-zeros, ~1400 dependency levels) against the oracle."""
    shape = (14, 12, 10)
    A = oracle.gen_fem(*shape)
    n = A.n_rows
    dA = ctx.gen_fem(*shape)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, n)
    dx, dy = ctx.upload(x), ctx.alloc(n)
    ctx.spmv(dA, dx, dy)
    assert relerr(dy.to_host(), oracle.spmv(A, x)) <= KTOL
    L, Ls, U, Us = oracle.split_LU(A)
    D, _, _ = oracle.peel_diag(L)
    dLs, dUs, dD, dDinv = ctx.split_strict(dA)
    ctx.sptrsv(dLs, dy, dD, dx)
    assert np.array_equal(dy.to_host(), oracle.sptrsv(Ls, D, x))
    ctx.bsptrsv(dUs, dy, dD, dx)
    assert np.array_equal(dy.to_host(), oracle.sptrsv(Us, D, x, backward=True))
    iLs, iL_D, iUs, iU_D = oracle.factor_ilu0(A)
    fLs, fL_D, fUs, fU_D = ctx.ilu0(dA)
    rp, col, val = fUs.download()
    assert np.array_equal(rp, iUs.row_ptr) and np.array_equal(col, iUs.col)
    assert relerr(val, iUs.val) <= KTOL
    rp, col, val = fLs.download()
    assert np.array_equal(col, iLs.col) and relerr(val, iLs.val) <= KTOL
    assert relerr(fU_D.to_host(), iU_D) <= KTOL


@pytest.mark.parametrize("name", GOLDEN_MATS)
def test_split_strict_bit_exact(ctx, name):
    """Device split_LU/peel_diag (LU_factors.hpp:122-309, :827-869) vs the
    reference's outputs."""
    g = load_golden(name)
    dA = ctx.matrix(crs_of(g, "A"))
    dLs, dUs, D, Dinv = ctx.split_strict(dA)
    for M, k in ((dLs, "Ls"), (dUs, "Us")):
        rp, col, val = M.download()
        assert np.array_equal(rp, g[k + "_rp"])
        assert np.array_equal(col, g[k + "_col"])
        assert np.array_equal(val, g[k + "_val"])
    assert np.array_equal(D.to_host(), g["A_D"])
    assert np.array_equal(Dinv.to_host(), g["A_D_inv"])


def test_split_strict_diag_errors(ctx):
    from basic_iterative_solvers_amd import BisError
    A = CRS(3, [0, 2, 3, 5], [0, 1, 0, 1, 2], [1.0, 2.0, 3.0, 4.0, 5.0])  # row 1 has no diagonal
    with pytest.raises(BisError, match="No diagonal to extract at row index 1"):
        ctx.split_strict(ctx.matrix(A))
    A = CRS(2, [0, 1, 2], [0, 1], [1.0, 0.0])
    with pytest.raises(BisError, match="Zero detected on diagonal at row index 1"):
        ctx.split_strict(ctx.matrix(A))


def test_spmv_ragged_unsorted_and_empty_rows(ctx, oracle):
    rng = np.random.default_rng(7)
    n = 5000
    lens = rng.integers(0, 40, n)
    lens[::97] = 0          # empty rows
    lens[1234] = 3000       # one long row inside the LDS budget
    rp = np.concatenate([[0], np.cumsum(lens)])
    col = rng.integers(0, n, rp[-1]).astype(np.int32)  # unsorted, duplicates allowed
    val = rng.uniform(-1, 1, rp[-1])
    A = CRS(n, rp, col, val)
    x = rng.uniform(-1, 1, n)
    y = dev_spmv(ctx, A, x)
    yo = oracle.spmv(A, x)
    scale = np.abs(A.to_scipy()).dot(np.abs(x)).max()
    assert np.max(np.abs(y - yo)) <= KTOL * scale


def test_spmv_very_long_rows_fallback(ctx, oracle):
    rng = np.random.default_rng(8)
    n = 64
    lens = np.full(n, 5)
    lens[3] = 20000  # exceeds the LDS budget -> wave-per-row kernel
    rp = np.concatenate([[0], np.cumsum(lens)])
    col = rng.integers(0, n, rp[-1]).astype(np.int32)
    val = rng.uniform(-1, 1, rp[-1])
    A = CRS(n, rp, col, val)
    x = rng.uniform(-1, 1, n)
    y = dev_spmv(ctx, A, x)
    yo = oracle.spmv(A, x)
    scale = np.abs(A.to_scipy()).dot(np.abs(x)).max()
    assert np.max(np.abs(y - yo)) <= KTOL * scale


def test_spmv_degenerate_shapes(ctx, oracle):
    # empty matrix, all-empty rows, single entry
    A = CRS(0, [0], [], [])
    dA = ctx.matrix(A)
    ctx.spmv(dA, ctx.alloc(1), ctx.alloc(1))
    A = CRS(4, [0, 0, 0, 0, 0], [], [])
    assert np.array_equal(dev_spmv(ctx, A, np.ones(4)), np.zeros(4))
    A = CRS(1, [0, 1], [0], [2.5])
    assert np.array_equal(dev_spmv(ctx, A, np.array([4.0])), np.array([10.0]))


def test_reference_unit_vectors(ctx):
    """tests/test_kernels.cpp:30-64, :72-88, :99-115 of the reference."""
    A = CRS(3, [0, 3, 6, 9], [0, 1, 2] * 3, np.arange(1.0, 10.0))
    assert np.allclose(dev_spmv(ctx, A, np.array([1.0, 2.0, 3.0])), [14, 32, 50], atol=1e-9)
    D = ctx.upload([2.0, 3.0, 4.0])
    Ls = ctx.matrix(CRS(3, [0, 0, 1, 3], [0, 0, 1], [1.0, -2.0, 1.0]))
    x = ctx.alloc(3)
    ctx.sptrsv(Ls, x, D, ctx.upload([2.0, 7.0, 12.0]))
    assert np.allclose(x.to_host(), [1, 2, 3], atol=1e-9)
    Us = ctx.matrix(CRS(3, [0, 2, 3, 3], [1, 2, 2], [1.0, -2.0, 1.0]))
    ctx.bsptrsv(Us, x, D, ctx.upload([-2.0, 9.0, 12.0]))
    assert np.allclose(x.to_host(), [1, 2, 3], atol=1e-9)
    assert ctx.euclidean_vec_norm(ctx.alloc(0), 0) == 0.0  # empty vector


@pytest.mark.parametrize("kind,size", [("hpcg", 32), ("anderson", 40)])
def test_medium_size_vs_oracle(ctx, oracle, kind, size):
    """Sizes the oracle finishes in seconds: SpMV, triangular solves (natural
    order arithmetic -> bit-exact against the fma oracle) and BLAS-1."""
    A = oracle.gen_hpcg(size) if kind == "hpcg" else oracle.gen_anderson(size, shift=9.0)
    n = A.n_rows
    rng = np.random.default_rng(12345)
    x, b = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    dA = ctx.gen_hpcg(size) if kind == "hpcg" else ctx.gen_anderson(size, shift=9.0)
    dx, db, out = ctx.upload(x), ctx.upload(b), ctx.alloc(n)
    ctx.spmv(dA, dx, out)
    yo = oracle.spmv(A, x)
    assert relerr(out.to_host(), yo) <= KTOL
    L, Ls, U, Us = oracle.split_LU(A)
    D, _, _ = oracle.peel_diag(L)
    dLs, dUs, dD, dDinv = ctx.split_strict(dA)
    assert np.array_equal(dD.to_host(), D)
    ctx.sptrsv(dLs, out, dD, db)
    assert np.array_equal(out.to_host(), oracle.sptrsv(Ls, D, b))
    ctx.bsptrsv(dUs, out, dD, db)
    assert np.array_equal(out.to_host(), oracle.sptrsv(Us, D, b, backward=True))
    assert abs(ctx.dot(dx, db) - oracle.dot(x, b)) <= KTOL * np.abs(x).dot(np.abs(b))
    assert abs(ctx.euclidean_vec_norm(dx) - oracle.norm(x)) <= KTOL * oracle.norm(x)


_H = load_histories()
_CG_KEYS = sorted(k for k in _H if k.split("|")[1] == "cg" and k.split("|")[2] in ("none", "j")
                  and "num_scale" not in k)


@pytest.mark.parametrize("key", _CG_KEYS)
def test_fused_cg_history_vs_reference(ctx, key):
    """bis_cg_* (fused device schedule) vs the reference's residual tables:
    max_k |r_k - r_k^ref| <= 1e-10 r_0, same iteration count."""
    e = _H[key]
    name, solver, pc, kw = parse_hist_key(key)
    g = load_golden(name)
    A = crs_of(g, "A")
    n = A.n_rows
    dA = ctx.matrix(A)
    b, x = ctx.upload(np.full(n, 1.0)), ctx.upload(np.full(n, 0.1))
    dD = ctx.upload(g["A_D"]) if pc == "j" else None
    cg = ctx.cg(dA, b, x, dD)
    r0 = cg.init(1e-14)
    assert abs(r0 - e["hist"][0]) <= 1e-13 * e["hist"][0]
    cg.iterate(1000)
    iters, conv, hist = cg.status()
    check_history(dict(hist=hist, iters=iters, converged=conv), e, "cg")
    if e["converged"]:
        # true residual of the returned x (solver.hpp:153-159)
        res, tmp = ctx.alloc(n), ctx.alloc(n)
        ctx.compute_residual(dA, x, b, res, tmp)
        assert ctx.euclidean_vec_norm(res) <= 1e-9 * e["hist"][0]
    cg.free()


def unfused_cg_history(ctx, dA, b, x, iters, D=None):
    """The reference's CG iteration (methods/cg.hpp:6-54, residual sample :162-166) call by call
    through the kernel entry points, host scalars as in the reference: the unfused schedule."""
    n = dA.n_rows
    r, z, p, tmp = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    ctx.compute_residual(dA, x, b, r, tmp)                    # cg.hpp:100-118
    ctx.apply_preconditioner("j" if D is not None else "none", n, None, None, D, None, None, None, z, r, tmp, tmp)
    ctx.copy_vector(p, z)
    hist = [ctx.euclidean_vec_norm(r)]
    for _ in range(iters):
        ctx.spmv(dA, p, tmp)                                  # :16
        rz = ctx.dot(r, z)                                    # :19
        alpha = rz / ctx.dot(tmp, p)                          # :23
        ctx.sum_vectors(x, x, p, alpha)                       # :28
        ctx.subtract_vectors(r, r, tmp, alpha)                # :31
        ctx.apply_preconditioner("j" if D is not None else "none", n, None, None, D, None, None, None, z, r, tmp, tmp)
        beta = ctx.dot(r, z) / rz                             # :47
        ctx.sum_vectors(p, z, p, beta)                        # :52
        hist.append(ctx.euclidean_vec_norm(r))                # :162-166
    for v in (r, z, p, tmp):
        v.free()
    return np.array(hist)


def hpcg_full_size_properties(ctx, n1, cg_iters):
    """Size-independent properties of the HPCG operator and of the CG built on it, for sizes no CPU
    oracle finishes (and, at 512^3, no reference CRS can hold): row sums and ||b - A x0|| in closed
    form, linearity, symmetry, and the fused CG schedule against the unfused kernels."""
    dA = ctx.gen_hpcg(n1)
    N = n1 ** 3
    assert dA.nnz == (3 * n1 - 2) ** 3
    assert dA.rp_width == (8 if dA.nnz >= 2 ** 31 - 8 else 4)
    ones, y = ctx.alloc(N), ctx.alloc(N)
    ctx.init_vector(ones, 1.0)
    ctx.spmv(dA, ones, y)
    # A*1 = 26 - (neighbours) = 27 - cx*cy*cz
    c = np.full(n1, 3.0); c[0] = c[-1] = 2.0
    expect = 27.0 - c[:, None, None] * c[None, :, None] * c[None, None, :]
    assert np.array_equal(y.to_host().reshape(n1, n1, n1), expect)
    # r0 = b - A x0 with b = 1, x0 = 0.1 (solver.hpp:101-102): 1 - 0.1*(27 - cx cy cz) per row
    vals, counts = np.unique(expect, return_counts=True)
    r0_closed = float(np.sqrt(np.sum(counts * (1.0 - 0.1 * vals) ** 2)))
    del expect
    rng = np.random.default_rng(1)
    du, dv, dw = ctx.upload(rng.uniform(-1, 1, N)), ctx.upload(rng.uniform(-1, 1, N)), ctx.alloc(N)
    yu, yv = ctx.alloc(N), ctx.alloc(N)
    ctx.sum_vectors(dw, du, dv, -0.75)
    ctx.spmv(dA, du, yu); ctx.spmv(dA, dv, yv); ctx.spmv(dA, dw, y)
    ctx.sum_vectors(ones, yu, yv, -0.75)
    ctx.subtract_vectors(ones, ones, y, 1.0)
    assert ctx.euclidean_vec_norm(ones) <= 1e-13 * 52 * np.sqrt(N)
    # the library's default stream format for this matrix (round 4: 32 bits per row, the row-mask kernel) against the kernel that
    # streams the CRS value array, on a second copy of the operator: y bit for bit
    assert dA.spmv_stream_info()[:2] == (0, 0) and dA.spmv_stream_info()[3] == 4
    # ... both kernels that stream 8-byte values: the window + sliced-ELL form (win8, the default without a dictionary) and the
    # row-block kernel on the CRS arrays in place (spmv_win8 = 0)
    for w8, form in ((-1, 6), (0, 0)):
        ctx.set_option("spmv_valdict", 0)
        ctx.set_option("spmv_win8", w8)
        try:
            dB = ctx.gen_hpcg(n1)
            ctx.spmv(dB, du, y)
            assert dB.spmv_stream_info()[1:] == (8, 0, form)
        finally:
            ctx.set_option("spmv_valdict", -1)
            ctx.set_option("spmv_win8", -1)
        ctx.subtract_vectors(y, y, yu, 1.0)
        assert ctx.euclidean_vec_norm(y) == 0.0
        dB.free()
    # symmetry: (Au, v) == (u, Av)
    a, b2 = ctx.dot(yu, dv), ctx.dot(du, yv)
    assert abs(a - b2) <= 1e-12 * max(abs(a), 1.0) * 10
    for v in (du, dv, dw, yu, yv):
        v.free()
    # fused CG (bis_cg_*: 3 passes, device scalars) against the unfused kernels, same x0 and b
    b, x = ones, y
    ctx.init_vector(b, 1.0)
    ctx.init_vector(x, 0.1)
    cg = ctx.cg(dA, b, x)
    r0 = cg.init(0.0)
    assert abs(r0 - r0_closed) <= 1e-12 * r0_closed
    cg.iterate(cg_iters)
    iters, conv, hist = cg.status()
    assert iters == cg_iters
    x_fused = x.to_host()
    cg.free()
    ctx.init_vector(x, 0.1)
    ref_hist = unfused_cg_history(ctx, dA, b, x, cg_iters)
    assert abs(ref_hist[0] - r0_closed) <= 1e-12 * r0_closed
    assert np.max(np.abs(hist - ref_hist)) <= 1e-10 * r0_closed
    xu = x.to_host()
    assert np.max(np.abs(x_fused - xu)) <= 1e-10 * np.max(np.abs(xu))
    dA.free(); b.free(); x.free()


def test_full_size_hpcg256_properties(ctx):
    """BASELINE metric size (HPCG 256^3, 449,455,096 nnz): size-independent
    properties -- row sums and ||r_0|| in closed form, linearity, symmetry, and the fused CG
    history staying consistent with the unfused kernels (1e-10 r_0 over 25 iterations)."""
    hpcg_full_size_properties(ctx, 256, 25)


def test_full_size_hpcg512_target_properties(ctx):
    """North-star target size (HPCG 512^3: 134,217,728 rows, 3,609,741,304 nnz > 2^31 -- int64
    row pointers; the reference's CRS cannot hold it, so there is no oracle by construction):
    the same size-independent gate on the RP = int64_t kernels at full size."""
    hpcg_full_size_properties(ctx, 512, 12)


def test_sptrsv_few_level_path_bit_exact(ctx, oracle):
    """Orderings with few, wide dependency levels (multi-colour) take the
    per-level streaming path (SpMV kernel + triangular epilogue: products are
    rounded before the left-to-right row sum, so <= 1e-13 instead of bit-exact);
    x aliases b."""
    rng = np.random.default_rng(3)
    n = 20000
    # red-black-like: rows [n/2, n) depend only on rows [0, n/2)
    lens = np.concatenate([np.zeros(n // 2, dtype=np.int64), rng.integers(1, 9, n - n // 2)])
    rp = np.concatenate([[0], np.cumsum(lens)])
    col = rng.integers(0, n // 2, rp[-1]).astype(np.int32)
    Ls = CRS(n, rp, col, rng.uniform(-1, 1, rp[-1]))
    D, b = rng.uniform(1, 2, n), rng.uniform(-1, 1, n)
    dLs, dD = ctx.matrix(Ls), ctx.upload(D)
    x = ctx.upload(b)
    ctx.sptrsv(dLs, x, dD, x)
    assert relerr(x.to_host(), oracle.sptrsv(Ls, D, b.copy())) <= KTOL
    # transposed structure -> backward solve
    Us = CRS(n, np.concatenate([rp[n // 2:] - rp[n // 2], np.full(n // 2, rp[-1])]),
             (col + n // 2).astype(np.int32), Ls.val)
    dUs = ctx.matrix(Us)
    x2 = ctx.alloc(n)
    ctx.bsptrsv(dUs, x2, dD, ctx.upload(b))
    assert relerr(x2.to_host(), oracle.sptrsv(Us, D, b.copy(), backward=True)) <= KTOL


@pytest.mark.parametrize("name", GOLDEN_MATS)
def test_device_ilu0_vs_reference_factors(ctx, name):
    """bis_mat_ilu0 (level-parallel on the device) vs the reference's serial
    factor_ILU0_old: same pattern (ascending columns), values <= 1e-13."""
    g = load_golden(name)
    dA = ctx.matrix(crs_of(g, "A"))
    dLs, L_D, dUs, U_D = ctx.ilu0(dA)
    rp, col, val = dLs.download()
    assert np.array_equal(rp, g["iluLs_rp"]) and np.array_equal(col, g["iluLs_col"])
    assert relerr(val, g["iluLs_val"]) <= KTOL
    rp, col, val = dUs.download()
    assert np.array_equal(rp, g["iluUs_rp"]) and np.array_equal(col, g["iluUs_col"])
    assert relerr(val, g["iluUs_val"]) <= KTOL
    assert relerr(U_D.to_host(), g["iluUD"]) <= KTOL
    assert np.array_equal(L_D.to_host(), g["iluLD"])


def test_device_ilu0_medium_vs_oracle(ctx, oracle):
    A = oracle.gen_hpcg(20)
    Ls, L_D, Us, U_D = oracle.factor_ilu0(A)
    dLs, dL_D, dUs, dU_D = ctx.ilu0(ctx.gen_hpcg(20))
    rp, col, val = dUs.download()
    assert np.array_equal(rp, Us.row_ptr) and np.array_equal(col, Us.col)
    assert relerr(val, Us.val) <= KTOL
    rp, col, val = dLs.download()
    assert np.array_equal(col, Ls.col) and relerr(val, Ls.val) <= KTOL
    assert relerr(dU_D.to_host(), U_D) <= KTOL


@pytest.mark.parametrize("kind", ["hpcg", "anderson", "fem", "klein"])
def test_device_multicolour_reordering(ctx, oracle, kind):
    """bis_mat_multicolour: a proper colouring (coupled rows never share a colour
    block), perm is a permutation grouped by colour, B == P A P^T entry for entry
    (row-internal order kept), and the colouring equals the sequential greedy
    first-fit in natural order (2 colours for the 7-point, 8 for the 27-point stencil)."""
    if kind == "klein":
        A = crs_of(load_golden("matrix_band_klein"), "A")
    else:
        A = {"hpcg": lambda: oracle.gen_hpcg(9, 7, 8), "anderson": lambda: oracle.gen_anderson(8, shift=9.0),
             "fem": lambda: oracle.gen_fem(7, 6, 5)}[kind]()
    n = A.n_rows
    dB, perm, n_col = ctx.multicolour(ctx.matrix(A))
    assert sorted(perm.tolist()) == list(range(n))
    # sequential greedy reference
    colour = np.full(n, -1)
    for r in range(n):
        used = {colour[c] for c in A.col[A.row_ptr[r]:A.row_ptr[r + 1]] if c < r}
        c = 0
        while c in used:
            c += 1
        colour[r] = c
    assert n_col == colour.max() + 1
    want = np.argsort(colour, kind="stable")
    assert np.array_equal(perm, want)
    if kind in ("hpcg", "anderson"):
        assert n_col == (8 if kind == "hpcg" else 2)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    rp, col, val = dB.download()
    lens = np.diff(A.row_ptr)[perm]
    assert np.array_equal(rp, np.concatenate([[0], np.cumsum(lens)]))
    wcol = np.concatenate([inv[A.col[A.row_ptr[o]:A.row_ptr[o + 1]]] for o in perm])
    wval = np.concatenate([A.val[A.row_ptr[o]:A.row_ptr[o + 1]] for o in perm])
    assert np.array_equal(col, wcol) and np.array_equal(val, wval)
    # the permuted operator still multiplies correctly (packed stream rebuilt for B)
    x = np.random.default_rng(3).uniform(-1, 1, n)
    dx, dy = ctx.upload(x[perm]), ctx.alloc(n)
    ctx.spmv(dB, dx, dy)
    assert relerr(dy.to_host(), oracle.spmv(A, x)[perm]) <= KTOL


@pytest.mark.parametrize("name", GOLDEN_MATS)
def test_device_scale_sym_bit_exact(ctx, oracle, name):
    """-scale on the device (bis_mat_scale_sym) against extract_scale + scale_mat of the oracle:
    scale vector and scaled values bit for bit; the SpMV on the scaled matrix (packed stream kept) agrees."""
    A = crs_of(load_golden(name), "A")
    dA = ctx.matrix(A)
    s_dev = ctx.scale_sym(dA)
    B = CRS(A.n_rows, A.row_ptr.copy(), A.col.copy(), A.val.copy())
    s_ref, st = oracle.extract_scale(B)
    assert st == 0
    oracle.scale_mat(B, s_ref)
    assert np.array_equal(s_dev.to_host(), s_ref)
    rp, col, val = dA.download()
    assert np.array_equal(val, B.val) and np.array_equal(col, B.col)
    x = np.random.default_rng(2).uniform(-1, 1, A.n_rows)
    dx, dy = ctx.upload(x), ctx.alloc(A.n_rows)
    ctx.spmv(dA, dx, dy)
    assert relerr(dy.to_host(), oracle.spmv(B, x)) <= KTOL


def test_vec_gather(ctx):
    rng = np.random.default_rng(21)
    n = 1000
    v = rng.uniform(-1, 1, n)
    perm = rng.permutation(n).astype(np.int32)
    dv, out = ctx.upload(v), ctx.alloc(n)
    ctx.gather(out, dv, perm)
    assert np.array_equal(out.to_host(), v[perm])


def test_tune_placement_keeps_the_matrix_intact(ctx, oracle):
    """bis_mat_tune_placement re-allocates the streamed arrays: same SpMV result bit for bit, same
    CRS download, for the packed stream and for the 32-bit column fallback (value dictionary off: the kernels that
    stream the re-allocated arrays), and with the dictionary, where the tuning leaves the matrix as it is."""
    from basic_iterative_solvers_amd import load_library
    lib = load_library()
    A = oracle.gen_hpcg(24)
    x = np.random.default_rng(11).uniform(-1, 1, A.n_rows)
    for packed, valdict in ((1, 0), (0, 0), (1, -1)):
        lib.bis_set_option(b"spmv_packed", packed)
        lib.bis_set_option(b"spmv_valdict", valdict)
        dA = ctx.matrix(A)
        dx, dy = ctx.upload(x), ctx.alloc(A.n_rows)
        ctx.spmv(dA, dx, dy)
        before = dy.to_host()
        first, best = ctx.tune_placement(dA, 3)
        assert 0 < best <= first
        ctx.spmv(dA, dx, dy)
        assert np.array_equal(dy.to_host(), before)
        rp, col, val = dA.download()
        assert np.array_equal(col, A.col) and np.array_equal(val, A.val)
        dA.free()
    lib.bis_set_option(b"spmv_packed", -1)
    lib.bis_set_option(b"spmv_valdict", -1)


def test_named_kernel_protocol(ctx, oracle):
    """The reference's plugin protocol (smax_helpers.hpp:7-42, kernels.hpp:48,
    cg.hpp:136-152, jacobi.hpp:93): register, run by name with an operand
    offset (gmres.hpp:168-169), swap_operands, rebind."""
    import ctypes as C
    lib = ctx.lib
    A = oracle.gen_hpcg(6)
    n = A.n_rows
    dA = ctx.matrix(A)
    rng = np.random.default_rng(2)
    V = rng.uniform(-1, 1, 3 * n)          # 3 basis vectors, like GMRES' V
    dV, dw = ctx.upload(V), ctx.alloc(n)
    name = b"w_j <- A*v_j"
    ctx.check(lib.bis_register_kernel(ctx.h, name, 0))
    ctx.check(lib.bis_kernel_register_A(ctx.h, name, dA.h))
    ctx.check(lib.bis_kernel_register_B(ctx.h, name, C.c_int64(3 * n), C.c_void_p(dV.ptr)))
    ctx.check(lib.bis_kernel_register_C(ctx.h, name, C.c_int64(n), C.c_void_p(dw.ptr)))
    for j in range(3):
        ctx.check(lib.bis_kernel_run(ctx.h, name, C.c_int64(0), C.c_int64(j * n), C.c_int64(0)))
        assert relerr(dw.to_host(), oracle.spmv(A, V[j * n:(j + 1) * n])) <= KTOL
    assert lib.bis_kernel_run(ctx.h, name, C.c_int64(0), C.c_int64(3 * n), C.c_int64(0)) != 0  # out of range
    assert lib.bis_kernel_run(ctx.h, b"no such kernel", C.c_int64(0), C.c_int64(0), C.c_int64(0)) != 0
    # Jacobi-style ping-pong: x_new <- A x_old, then swap_operands
    x0, x1 = ctx.upload(V[:n]), ctx.alloc(n)
    ctx.check(lib.bis_register_kernel(ctx.h, b"x_new <- A*x_old", 0))
    ctx.check(lib.bis_kernel_register_A(ctx.h, b"x_new <- A*x_old", dA.h))
    ctx.check(lib.bis_kernel_register_B(ctx.h, b"x_new <- A*x_old", C.c_int64(n), C.c_void_p(x0.ptr)))
    ctx.check(lib.bis_kernel_register_C(ctx.h, b"x_new <- A*x_old", C.c_int64(n), C.c_void_p(x1.ptr)))
    ctx.check(lib.bis_kernel_run(ctx.h, b"x_new <- A*x_old", C.c_int64(0), C.c_int64(0), C.c_int64(0)))
    ctx.check(lib.bis_kernel_swap_operands(ctx.h, b"x_new <- A*x_old"))
    ctx.check(lib.bis_kernel_run(ctx.h, b"x_new <- A*x_old", C.c_int64(0), C.c_int64(0), C.c_int64(0)))
    assert relerr(x0.to_host(), oracle.spmv(A, oracle.spmv(A, V[:n]))) <= KTOL
    # triangular solves by name, upper flag
    L, Ls, U, Us = oracle.split_LU(A)
    D, _, _ = oracle.peel_diag(L)
    dLs, dUs, dD = ctx.matrix(Ls), ctx.matrix(Us), ctx.upload(D)
    b = rng.uniform(-1, 1, n)
    db, dx = ctx.upload(b), ctx.alloc(n)
    for nm, M, up, ref in ((b"solve L", dLs, 0, oracle.sptrsv(Ls, D, b)),
                           (b"solve U", dUs, 1, oracle.sptrsv(Us, D, b, backward=True))):
        ctx.check(lib.bis_register_kernel(ctx.h, nm, 1))
        ctx.check(lib.bis_kernel_register_A(ctx.h, nm, M.h))
        ctx.check(lib.bis_kernel_register_B(ctx.h, nm, C.c_int64(n), C.c_void_p(dx.ptr)))
        ctx.check(lib.bis_kernel_register_C(ctx.h, nm, C.c_int64(n), C.c_void_p(db.ptr)))
        ctx.check(lib.bis_kernel_register_D(ctx.h, nm, C.c_void_p(dD.ptr)))
        ctx.check(lib.bis_kernel_set_mat_upper_triang(ctx.h, nm, up))
        ctx.check(lib.bis_kernel_run(ctx.h, nm, C.c_int64(0), C.c_int64(0), C.c_int64(0)))
        assert np.array_equal(dx.to_host(), ref)


def test_full_size_fem_config5_properties(ctx):
    """BASELINE config 5 size (stand-in for Flan_1565: 80x80x81 nodes, 1.56 M rows,
    ~1.0e8 non-zeros): symmetry of the operator; triangular solves of the device
    ILU(0) inverted exactly by a product; and (L U)_ij == A_ij on A's pattern for
    sampled rows (the defining property of ILU(0))."""
    import scipy.sparse as sp
    shape = (80, 80, 81)
    dA = ctx.gen_fem(*shape)
    N = dA.n_rows
    assert N == 3 * 80 * 80 * 81 and 60 * N < dA.nnz < 81 * N
    rng = np.random.default_rng(6)
    u, v = rng.uniform(-1, 1, N), rng.uniform(-1, 1, N)
    du, dv, yu, yv = ctx.upload(u), ctx.upload(v), ctx.alloc(N), ctx.alloc(N)
    ctx.spmv(dA, du, yu); ctx.spmv(dA, dv, yv)
    a, b2 = ctx.dot(yu, dv), ctx.dot(du, yv)
    assert abs(a - b2) <= 1e-12 * max(abs(a), abs(b2), 1.0)
    fLs, fL_D, fUs, fU_D = ctx.ilu0(dA)
    x, t = ctx.alloc(N), ctx.alloc(N)
    for solve, T, D in ((ctx.sptrsv, fLs, fL_D), (ctx.bsptrsv, fUs, fU_D)):
        solve(T, x, D, du)
        ctx.spmv(T, x, t)                       # t = T x
        ctx.elemwise_mult_vectors(yu, D, x)     # yu = D x
        ctx.sum_vectors(t, t, yu)               # (D + T) x
        ctx.subtract_vectors(t, t, du)
        assert ctx.euclidean_vec_norm(t) <= 1e-12 * np.sqrt(N) * 20
    rows = np.sort(rng.choice(N, 400, replace=False))
    arp, acol, aval = dA.download()
    lrp, lcol, lval = fLs.download()
    urp, ucol, uval = fUs.download()
    A = sp.csr_matrix((aval, acol, arp), shape=(N, N))
    L = sp.csr_matrix((lval, lcol, lrp), shape=(N, N)) + sp.identity(N, format="csr")
    U = sp.csr_matrix((uval, ucol, urp), shape=(N, N)) + sp.diags(fU_D.to_host(), format="csr")
    P = (L[rows] @ U).tocsr()
    Ar = A[rows].tocsr()
    worst = 0.0
    for i in range(len(rows)):
        cols = Ar.indices[Ar.indptr[i]:Ar.indptr[i + 1]]
        want = Ar.data[Ar.indptr[i]:Ar.indptr[i + 1]]
        got = np.asarray(P[i, cols].todense()).ravel()
        worst = max(worst, np.abs(got - want).max())
    assert worst <= 1e-12


def test_full_size_anderson256_properties(ctx):
    """BASELINE configs 2-4 size (Anderson 256^3, 16.7M rows): symmetry of the
    operator, and the triangular solves inverted exactly by a product:
    (D + L) sptrsv(L, D, b) == b and (D + U) bsptrsv(U, D, b) == b."""
    L1 = 256
    N = L1 ** 3
    dA = ctx.gen_anderson(L1, shift=9.0)
    assert dA.nnz == 7 * N
    rng = np.random.default_rng(4)
    u, v = rng.uniform(-1, 1, N), rng.uniform(-1, 1, N)
    du, dv, yu, yv = ctx.upload(u), ctx.upload(v), ctx.alloc(N), ctx.alloc(N)
    ctx.spmv(dA, du, yu); ctx.spmv(dA, dv, yv)
    a, b2 = ctx.dot(yu, dv), ctx.dot(du, yv)
    assert abs(a - b2) <= 1e-12 * max(abs(a), abs(b2), 1.0)
    dLs, dUs, dD, dDinv = ctx.split_strict(dA)
    x, t = ctx.alloc(N), ctx.alloc(N)
    for solve, T in ((ctx.sptrsv, dLs), (ctx.bsptrsv, dUs)):
        solve(T, x, dD, du)
        ctx.spmv(T, x, t)                       # t = T x
        ctx.elemwise_mult_vectors(yu, dD, x)    # yu = D x
        ctx.sum_vectors(t, t, yu)               # (D + T) x
        ctx.subtract_vectors(t, t, du)
        assert ctx.euclidean_vec_norm(t) <= 1e-13 * np.sqrt(N) * 20


@pytest.mark.parametrize("kind,size", [("anderson", 24), ("hpcg", 12)])
def test_sptrsv_lost_handoff_is_reported(ctx, oracle, kind, size):
    """A hand-off that never arrives in the natural-order sweeps (lane-per-row kernel: 3 dependencies per
    row; wave-per-row kernel: 13) must not hang and must not return NaNs silently: the waiting row gives
    up after its bounded spin, the sweep finishes, and the next blocking call fails with BIS_ERR_SYNC.
    Injected through the test hook `trsv_inject_loss` (one dependency of one row is redirected, for one
    sweep, to a scratch slot nobody publishes).  The sweep after it is bit-exact again."""
    from basic_iterative_solvers_amd import BisError
    A = oracle.gen_hpcg(size) if kind == "hpcg" else oracle.gen_anderson(size, shift=9.0)
    n = A.n_rows
    L, Ls, U, Us = oracle.split_LU(A)
    D, _, _ = oracle.peel_diag(L)
    dA = ctx.gen_hpcg(size) if kind == "hpcg" else ctx.gen_anderson(size, shift=9.0)
    dLs, dUs, dD, dDinv = ctx.split_strict(dA)
    b = np.random.default_rng(9).uniform(-1, 1, n)
    db, x = ctx.upload(b), ctx.alloc(n)
    want = oracle.sptrsv(Ls, D, b)
    ctx.sptrsv(dLs, x, dD, db)               # builds the plan and the position table
    assert np.array_equal(x.to_host(), want)
    ctx.set_option("trsv_inject_loss", Ls.nnz // 2)
    ctx.sptrsv(dLs, x, dD, db)               # (builds the level plan: the hook lives in the level-scheduled sweeps)
    with pytest.raises(BisError, match="lost a hand-off"):
        ctx.sync()
    ctx.sync()                               # reported once
    ctx.sptrsv(dLs, x, dD, db)
    assert np.array_equal(x.to_host(), want)
    # wall time of a failing sweep: the first wait gives up after its bounded spin (about a second), every other wait
    # sees the fault word within a millisecond and later rows do not wait at all -- the grid drains at once, it does
    # not give up row by row.  Hundreds of rows depend on the lost one here.
    import time
    ctx.sync()
    ctx.set_option("trsv_tiled", 0)
    try:
        ctx.sptrsv(dLs, x, dD, db)
        ctx.sync()
        ctx.set_option("trsv_inject_loss", Ls.nnz // 8)
        t0 = time.perf_counter()
        ctx.sptrsv(dLs, x, dD, db)
        with pytest.raises(BisError, match="lost a hand-off"):
            ctx.sync()
        assert time.perf_counter() - t0 < 10.0
        ctx.sptrsv(dLs, x, dD, db)
        assert np.array_equal(x.to_host(), want)
    finally:
        ctx.set_option("trsv_tiled", -1)


@pytest.mark.parametrize("kind", ["j", "gs", "sgs"])
@pytest.mark.parametrize("mat", ["hpcg10", "anderson9", "hpcg24"])
def test_stationary_device_schedules(ctx, oracle, kind, mat):
    """bis_stat_*: Jacobi / GS / SGS with the iteration, the true residual of every iterate, its norm and the
    stopping test on the device (jacobi.hpp:43-52,:102-107; gauss_seidel.hpp:26-52,:99-104; solver.hpp:177-192).
    Against the same iteration made kernel by kernel through the C ABI with a blocking norm (the reference's
    schedule): sampled norms bit-identical, same stopping iteration, same iterate; against the oracle's solver:
    history within 1e-10 r0, same iteration count; launches enqueued behind the stopping iteration change nothing."""
    A = {"hpcg10": lambda: oracle.gen_hpcg(10), "anderson9": lambda: oracle.gen_anderson(9, shift=9.0), "hpcg24": lambda: oracle.gen_hpcg(24)}[mat]()
    dA = {"hpcg10": lambda: ctx.gen_hpcg(10), "anderson9": lambda: ctx.gen_anderson(9, shift=9.0), "hpcg24": lambda: ctx.gen_hpcg(24)}[mat]()
    n = A.n_rows
    tol, max_iters = 1e-14, 400
    dLs, dUs, dD, dDinv = ctx.split_strict(dA)
    b = ctx.upload(np.full(n, 1.0))
    # the kernel-by-kernel schedule
    x, xn, t, r = ctx.upload(np.full(n, 0.1)), ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    ctx.compute_residual(dA, x, b, r, t)
    hist = [ctx.euclidean_vec_norm(r)]
    for it in range(max_iters):
        if kind == "j":
            ctx.spmv(dA, x, xn)
            ctx.normalize_x(xn, x, dD, b)
            x, xn = xn, x
        else:
            ctx.spmv(dUs, x, t); ctx.subtract_vectors(t, b, t); ctx.sptrsv(dLs, x, dD, t)
            if kind == "sgs":
                ctx.spmv(dLs, x, t); ctx.subtract_vectors(t, b, t); ctx.bsptrsv(dUs, x, dD, t)
        ctx.compute_residual(dA, x, b, r, t)
        hist.append(ctx.euclidean_vec_norm(r))
        if hist[-1] < tol * hist[0] or not np.isfinite(hist[-1]):
            break
    hist = np.array(hist)
    x_sep = x.to_host()
    # the device schedule, enqueued in one go and well past the stopping iteration
    y = ctx.upload(np.full(n, 0.1))
    st = ctx.stat(kind, dA, dD, b, y, dLs, dUs)
    r0 = st.init(tol)
    st.iterate(7)
    st.iterate(max_iters - 7)
    iters, conv, h2 = st.status(hist_cap=1024)
    out = ctx.alloc(n)
    st.solution(out)
    if conv:  # launches behind the stopping iteration are no-ops
        st.iterate(25)
        it3, conv3, h3 = st.status(hist_cap=1024)
        out3 = ctx.alloc(n)
        st.solution(out3)
        assert it3 == iters and conv3 and np.array_equal(h3, h2) and np.array_equal(out3.to_host(), out.to_host())
    assert r0 == hist[0] and iters == len(hist) - 1 and conv == (hist[-1] < tol * hist[0])
    assert np.array_equal(h2, hist)
    assert np.array_equal(out.to_host(), x_sep)
    ref = oracle.solve(A, kind, "none", max_iters=max_iters, tol=tol)
    m = min(len(ref["hist"]), len(h2))
    assert np.max(np.abs(ref["hist"][:m] - h2[:m])) <= 1e-10 * ref["hist"][0]
    if conv:
        assert abs(iters - ref["iters"]) <= 1
    st.free()


def test_tiled_plan_out_of_memory_is_not_an_error(ctx, oracle, capfd, monkeypatch):
    """The tiled sweep's plan is an optimisation with sizeable scratch (two hash tables of 4 E words): when the device
    cannot hold it (test hook `trsv_inject_oom`) the solve must not fail -- the plan is 'not applicable', the error state
    is cleared and the level-scheduled sweep computes the same bits (kernels.hpp:54-107)."""
    A = oracle.gen_hpcg(20)
    n = A.n_rows
    L, Ls, U, Us = oracle.split_LU(A)
    D, _, _ = oracle.peel_diag(L)
    b = np.random.default_rng(31).uniform(-1, 1, n)
    monkeypatch.setenv("BIS_TRSV_TILE_STATS", "1")
    ctx.set_option("trsv_inject_oom", 1)
    try:
        dLs, dUs, dD, dDinv = ctx.split_strict(ctx.gen_hpcg(20))
        db, x = ctx.upload(b), ctx.alloc(n)
        capfd.readouterr()
        ctx.sptrsv(dLs, x, dD, db)
        assert np.array_equal(x.to_host(), oracle.sptrsv(Ls, D, b))
        ctx.bsptrsv(dUs, x, dD, db)
        assert np.array_equal(x.to_host(), oracle.sptrsv(Us, D, b, backward=True))
        ctx.sync()
        assert not [l for l in capfd.readouterr().err.splitlines() if l.startswith("tiled sptrsv plan")]  # no tiled plan was made
    finally:
        ctx.set_option("trsv_inject_oom", -1)
    ctx.sptrsv(dLs, x, dD, db)  # the matrix keeps its level-scheduled sweep
    assert np.array_equal(x.to_host(), oracle.sptrsv(Ls, D, b))


def test_sweep_plans_follow_values_changed_in_place(ctx, oracle):
    """The tiled sweep's plan holds a copy of the triangle's values in its entry stream, row views of the level plans
    hold dictionaries: after the values change in place (written through bis_mat_debug_ptrs, then bis_mat_retune --
    bis_mat_scale_sym takes the same path) the next sweep must use the new values, bit-exact against the oracle
    (kernels.hpp:54-107), with the tiled and with the level-scheduled kernels."""
    from basic_iterative_solvers_amd import Vec
    A = oracle.gen_hpcg(20)
    n = A.n_rows
    L, Ls, U, Us = oracle.split_LU(A)
    D, _, _ = oracle.peel_diag(L)
    b = np.random.default_rng(21).uniform(-1, 1, n)
    for tiled in (-1, 0):
        ctx.set_option("trsv_tiled", tiled)
        try:
            dLs, dUs, dD, dDinv = ctx.split_strict(ctx.gen_hpcg(20))
            db, x = ctx.upload(b), ctx.alloc(n)
            ctx.sptrsv(dLs, x, dD, db)
            assert np.array_equal(x.to_host(), oracle.sptrsv(Ls, D, b))
            new_val = Ls.val * np.random.default_rng(22).uniform(0.5, 1.5, Ls.nnz)
            Vec(ctx, dLs.debug_ptrs()[2], Ls.nnz, False).set(new_val)
            dLs.retune()
            ctx.sptrsv(dLs, x, dD, db)
            Ls2 = CRS(n, Ls.row_ptr, Ls.col, new_val)
            assert np.array_equal(x.to_host(), oracle.sptrsv(Ls2, D, b)), tiled
        finally:
            ctx.set_option("trsv_tiled", -1)


def test_sptrsv_wave_grid_is_capped_by_residency(ctx, oracle):
    """The wave-per-row sweep deals rows to waves round robin and needs its whole grid resident: a
    `trsv_grid` request far beyond what fits the device is capped by the occupancy query (not launched
    as asked), so the result stays bit-exact instead of waiting for workgroups that never start."""
    A = oracle.gen_fem(9, 8, 7)
    n = A.n_rows
    L, Ls, U, Us = oracle.split_LU(A)
    D, _, _ = oracle.peel_diag(L)
    dLs, dUs, dD, dDinv = ctx.split_strict(ctx.gen_fem(9, 8, 7))
    b = np.random.default_rng(10).uniform(-1, 1, n)
    db, x = ctx.upload(b), ctx.alloc(n)
    ctx.set_option("trsv_grid", 1 << 20)
    try:
        ctx.sptrsv(dLs, x, dD, db)
        assert np.array_equal(x.to_host(), oracle.sptrsv(Ls, D, b))
        ctx.bsptrsv(dUs, x, dD, db)
        assert np.array_equal(x.to_host(), oracle.sptrsv(Us, D, b, backward=True))
    finally:
        ctx.set_option("trsv_grid", -1)


def _tiled_cases():
    sizes = {"anderson": 40, "hpcg": 32, "fem": (14, 12, 10)}
    cases = []
    for kind, rows in (("anderson", 700), ("hpcg", 300), ("fem", 100)):  # interval tiles of the natural order: host plan
        cases += [(kind, sizes[kind], -1, 0, 1, 0), (kind, sizes[kind], rows, 0, 1, 0)]
    for kind, edge in (("anderson", 5), ("hpcg", 6), ("fem", 4), ("hpcg", 5 | 3 << 8 | 2 << 16)):  # grid tiles: host (2) and device (-1) plan
        for mode in (2, -1):
            cases += [(kind, sizes[kind], -1, -1, mode, 0), (kind, sizes[kind], -1, edge, mode, 0)]
    cases += [("hpcg", 32, -1, -1, -1, 1), ("fem", (14, 12, 10), -1, -1, -1, 1)]  # device plan over a 64-bit row_ptr
    return cases


@pytest.mark.parametrize("kind,size,rows,edge,mode,rp64", _tiled_cases())
def test_tiled_sweep_bit_exact(ctx, oracle, capfd, monkeypatch, kind, size, rows, edge, mode, rp64):
    """The tiled natural-order sweep (bis_trsv_tiled.hip: tiles solved by one workgroup each, in-tile operands
    through LDS rings, external ones through a poller wave): same CRS-order fma chain per row as the reference's
    serial loop -> bit-exact against the fma oracle, forward and backward, also with x aliasing b (gmres.hpp:173)
    and with tiles much smaller than the default (every ring wraps, operands leave the own-result ring and come back
    through the poller).  Its plan comes from the host (interval tiles, or grid tiles as the device plan's oracle) or
    from the device (grid tiles, the default on generated matrices): the two must describe the same plan."""
    A = {"anderson": lambda: oracle.gen_anderson(size, shift=9.0), "hpcg": lambda: oracle.gen_hpcg(size),
         "fem": lambda: oracle.gen_fem(*size)}[kind]()
    make = {"anderson": lambda: ctx.gen_anderson(size, shift=9.0), "hpcg": lambda: ctx.gen_hpcg(size),
            "fem": lambda: ctx.gen_fem(*size)}[kind]
    n = A.n_rows
    L, Ls, U, Us = oracle.split_LU(A)
    D, _, _ = oracle.peel_diag(L)
    b = np.random.default_rng(17).uniform(-1, 1, n)
    monkeypatch.setenv("BIS_TRSV_TILE_STATS", "1")
    ctx.set_option("trsv_tiled", mode)
    ctx.set_option("trsv_tile_rows", rows)
    ctx.set_option("trsv_tile_edge", edge)
    ctx.set_option("force_rp64", rp64)
    try:
        dA = make()
        dLs, dUs, dD, dDinv = ctx.split_strict(dA)
        assert dLs.rp_width == (8 if rp64 else 4)
        db, x = ctx.upload(b), ctx.alloc(n)
        want_f, want_b = oracle.sptrsv(Ls, D, b), oracle.sptrsv(Us, D, b, backward=True)
        capfd.readouterr()
        for _ in range(2):  # the second sweep reuses the plan and the scratch
            ctx.sptrsv(dLs, x, dD, db)
            assert np.array_equal(x.to_host(), want_f)
            ctx.bsptrsv(dUs, x, dD, db)
            assert np.array_equal(x.to_host(), want_b)
        x.set(b)
        ctx.sptrsv(dLs, x, dD, x)
        assert np.array_equal(x.to_host(), want_f)
        ctx.sync()
        lines = [l for l in capfd.readouterr().err.splitlines() if l.startswith("tiled sptrsv plan")]
        assert len(lines) == 2  # one plan per triangle, built once
        if edge == 0:
            assert all("interval tiles" in l for l in lines)
        else:
            assert all("grid tiles" in l and (("device plan" in l) == (mode == -1)) for l in lines)
        if mode == -1 and not rp64:  # the host plan of the same triangles: same tiles, steps, quads, external ordinals
            ctx.set_option("trsv_tiled", 2)
            hLs, hUs, _, _ = ctx.split_strict(dA)
            ctx.sptrsv(hLs, x, dD, db)
            ctx.bsptrsv(hUs, x, dD, db)
            ctx.sync()
            host = [l for l in capfd.readouterr().err.splitlines() if l.startswith("tiled sptrsv plan")]
            counts = lambda l: l.split(";")[0]
            assert [counts(l) for l in host] == [counts(l) for l in lines]
    finally:
        ctx.set_option("trsv_tiled", -1)
        ctx.set_option("trsv_tile_rows", -1)
        ctx.set_option("trsv_tile_edge", -1)
        ctx.set_option("force_rp64", -1)


def test_tiled_sweep_grid_hint(ctx, oracle, capfd, monkeypatch):
    """bis_mat_set_grid_hint on an uploaded matrix: a true hint gives the tiled sweep (device plan; pairs of x
    neighbours read as two unknowns of one node are a true description too), a hint that does not describe the
    pattern is found out by the plan's own checks (no admissible skew, or the order is not a linear extension) and
    costs nothing but falling back to the level-scheduled kernels; without a hint bis_mat_create recognises the grid
    from the column offsets of a few rows (option grid_autodetect); the sweeps are bit-exact in every case."""
    monkeypatch.setenv("BIS_TRSV_TILE_STATS", "1")
    A = oracle.gen_hpcg(24)
    n = A.n_rows
    L, Ls, U, Us = oracle.split_LU(A)
    D, _, _ = oracle.peel_diag(L)
    b = np.random.default_rng(23).uniform(-1, 1, n)
    want_f, want_b = oracle.sptrsv(Ls, D, b), oracle.sptrsv(Us, D, b, backward=True)
    db, x = ctx.upload(b), ctx.alloc(n)
    for hint, tiled in (((24, 24, 24, 1), True), ((12, 24, 24, 2), True), ((48, 12, 24, 1), False), ((16, 36, 24, 1), None), (None, True), ("off", False)):
        ctx.set_option("grid_autodetect", 0 if hint == "off" else -1)
        dA = ctx.matrix(A)  # no hint given: the grid is recognised from the rows' column offsets (unless switched off)
        ctx.set_option("grid_autodetect", -1)
        if hint and hint != "off":
            dA.set_grid_hint(*hint)
        dLs, dUs, dD, _ = ctx.split_strict(dA)
        capfd.readouterr()
        ctx.sptrsv(dLs, x, dD, db)
        assert np.array_equal(x.to_host(), want_f)
        ctx.bsptrsv(dUs, x, dD, db)
        assert np.array_equal(x.to_host(), want_b)
        ctx.sync()
        lines = [l for l in capfd.readouterr().err.splitlines() if l.startswith("tiled sptrsv plan")]
        assert tiled is None or len(lines) == (2 if tiled else 0), (hint, lines)
    with pytest.raises(Exception):
        ctx.matrix(A).set_grid_hint(24, 24, 23, 1)  # extents must multiply to the row count


def _queue_order(A, rcm):
    """The sequential definition of the two orderings (host/utilities/permute.hpp bfs_like_permutation)."""
    n = A.n_rows
    adj = [set() for _ in range(n)]
    for r in range(n):
        for c in A.col[A.row_ptr[r]:A.row_ptr[r + 1]]:
            if c != r:
                adj[r].add(int(c)); adj[int(c)].add(r)
    adj = [sorted(a) for a in adj]
    deg = [len(a) for a in adj]
    starts = sorted(range(n), key=lambda v: deg[v]) if rcm else list(range(n))
    seen, perm = [False] * n, []
    for s0 in starts:
        if seen[s0]:
            continue
        seen[s0] = True
        head = len(perm)
        perm.append(s0)
        while head < len(perm):
            v = perm[head]; head += 1
            nb = [w for w in adj[v] if not seen[w]]
            for w in nb:
                seen[w] = True
            if rcm:
                nb.sort(key=lambda w: deg[w])
            perm += nb
    return np.array(perm[::-1] if rcm else perm, dtype=np.int32)


@pytest.mark.parametrize("rcm", [False, True])
@pytest.mark.parametrize("kind", ["hpcg", "anderson", "fem", "klein", "components"])
def test_device_bfs_rcm_ordering(ctx, oracle, kind, rcm):
    """bis_mat_bfs_order: the level-synchronous device ordering equals the sequential queue algorithm entry for
    entry (also across several connected components), and bis_mat_permute builds P A P^T with the row-internal
    entry order kept."""
    if kind == "klein":
        A = crs_of(load_golden("matrix_band_klein"), "A")
    elif kind == "components":  # three disconnected 7-point grids of different sizes -> three BFS restarts
        import scipy.sparse as sp
        parts = [oracle.gen_anderson(L, shift=9.0).to_scipy() for L in (4, 3, 5)]
        M = sp.block_diag(parts, format="csr")
        A = CRS(M.shape[0], M.indptr, M.indices.astype(np.int32), M.data)
    else:
        A = {"hpcg": lambda: oracle.gen_hpcg(9, 7, 8), "anderson": lambda: oracle.gen_anderson(8, shift=9.0),
             "fem": lambda: oracle.gen_fem(7, 6, 5)}[kind]()
    n = A.n_rows
    dA = ctx.matrix(A)
    perm = ctx.bfs_order(dA, rcm)
    assert np.array_equal(perm, _queue_order(A, rcm))
    dB = ctx.permute(dA, perm)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    rp, col, val = dB.download()
    lens = np.diff(A.row_ptr)[perm]
    assert np.array_equal(rp, np.concatenate([[0], np.cumsum(lens)]))
    assert np.array_equal(col, np.concatenate([inv[A.col[A.row_ptr[o]:A.row_ptr[o + 1]]] for o in perm]))
    assert np.array_equal(val, np.concatenate([A.val[A.row_ptr[o]:A.row_ptr[o + 1]] for o in perm]))


def test_device_bfs_refuses_unsymmetric_patterns_and_bad_permutations(ctx):
    from basic_iterative_solvers_amd import BisError
    A = CRS(3, [0, 2, 3, 4], [0, 1, 1, 2], [1.0, 2.0, 3.0, 4.0])  # (0,1) without (1,0)
    dA = ctx.matrix(A)
    with pytest.raises(BisError, match="structurally symmetric"):
        ctx.bfs_order(dA)
    with pytest.raises(BisError, match="not a permutation"):
        ctx.permute(dA, np.array([0, 0, 2], dtype=np.int32))


def _crs_from_rows(rows):
    rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])])
    col = np.concatenate([np.asarray(r, dtype=np.int64) for r in rows]) if rp[-1] else np.zeros(0, np.int64)
    return CRS(len(rows), rp.astype(np.int32), col.astype(np.int32), np.ones(len(col)))


def test_device_symmetry_check_long_rows_and_compensating_defects(ctx):
    """The structure check of bis_mat_bfs_order (a wave per row: mirrors of the entries below the diagonal looked up 64
    columns at a time + a count of the entries on both sides): rows of several 64-entry chunks with columns in any order
    pass; a missing mirror in a late chunk, a missing mirror balanced by a stray entry elsewhere (the counts agree), and a
    column repeated inside a chunk or across two chunks are all refused."""
    from basic_iterative_solvers_amd import BisError
    rng = np.random.default_rng(3)
    n = 300
    dense = rng.random((n, n)) < 0.6
    dense = dense | dense.T
    np.fill_diagonal(dense, True)
    rows = [rng.permutation(np.flatnonzero(dense[r])).tolist() for r in range(n)]  # ~230 entries per row, unsorted
    A = _crs_from_rows(rows)
    dA = ctx.matrix(A)
    assert np.array_equal(ctx.bfs_order(dA, True), _queue_order(A, True))
    dA.free()

    def refused(rows2):
        d = ctx.matrix(_crs_from_rows(rows2))
        with pytest.raises(BisError, match="structurally symmetric"):
            ctx.bfs_order(d)
        d.free()

    # the mirror (5, 290) of a lower entry sitting in row 290's last chunk is gone
    r2 = [list(r) for r in rows]
    lower = [c for c in r2[290] if c < 290]
    victim = r2[290][-1] if r2[290][-1] < 290 else lower[-1]
    r2[victim] = [c for c in r2[victim] if c != 290]
    refused(r2)
    # ... and with a stray upper entry somewhere else the counts on both sides of the diagonal agree again
    free_col = next(c for c in range(n - 1, 0, -1) if c not in r2[0] and c != victim)
    r3 = [list(r) for r in r2]
    r3[0] = r3[0] + [free_col]
    refused(r3)
    # a column twice: inside one chunk, and in two different chunks of a long row
    r4 = [list(r) for r in rows]
    r4[7] = r4[7][:3] + [r4[7][1]] + r4[7][3:]
    refused(r4)
    r5 = [list(r) for r in rows]
    assert len(r5[200]) > 140
    r5[200] = r5[200] + [r5[200][2]]  # first chunk's column again in the last chunk
    refused(r5)
    # small: lower entry without a mirror, balanced by an upper entry without one
    refused([[0, 2], [0, 1], [2]])


_PCG_KEYS = sorted(k for k in _H if k.split("|")[1] == "cg" and k.split("|")[2] in ("sgs", "ilu0", "s2st", "j")
                   and "num_scale" not in k)


@pytest.mark.parametrize("key", _PCG_KEYS)
def test_fused_cg_general_preconditioner_vs_reference(ctx, key):
    """bis_cg_set_preconditioner: the fused device CG with z = M^-1 r through bis_apply_preconditioner (triangular
    sweeps of SGS / ILU(0), the two-stage form, Jacobi) against the reference's -cg -p ... residual tables."""
    e = _H[key]
    name, solver, pc, kw = parse_hist_key(key)
    g = load_golden(name)
    A = crs_of(g, "A")
    n = A.n_rows
    dA = ctx.matrix(A)
    b, x = ctx.upload(np.full(n, 1.0)), ctx.upload(np.full(n, 0.1))
    ones = ctx.upload(np.ones(n))
    if pc == "ilu0":
        Ls, L_D, Us, U_D = ctx.ilu0(dA)
        args = dict(Ls=Ls, Us=Us, A_D=ones, A_D_inv=ones, L_D=L_D, U_D=U_D)
    else:
        Ls, Us, D, Dinv = ctx.split_strict(dA)
        args = dict(Ls=Ls, Us=Us, A_D=D, A_D_inv=Dinv, L_D=ones, U_D=ones)
    cg = ctx.cg(dA, b, x)
    cg.set_preconditioner(pc, **args)
    r0 = cg.init(1e-14)
    assert abs(r0 - e["hist"][0]) <= 1e-13 * e["hist"][0]
    cg.iterate(1000)
    iters, conv, hist = cg.status()
    check_history(dict(hist=hist, iters=iters, converged=conv), e, "cg")
    cg.free()


def _few_values_matrix(rng, n, n_values, empty_head=0):
    """Random pattern (ragged, unsorted, duplicates allowed, `empty_head` leading empty rows) whose values are drawn
    from `n_values` distinct bit patterns, among them -0.0, 0.0, a denormal and both infinities' neighbours."""
    lens = rng.integers(0, 40, n)
    lens[:empty_head] = 0
    lens[::53] = 0
    rp = np.concatenate([[0], np.cumsum(lens)])
    col = rng.integers(0, n, rp[-1]).astype(np.int32)
    special = np.array([-0.0, 0.0, 5e-324, -1.0, 26.0, 1.7976931348623157e308, -1.7976931348623157e308, 1.0 + 2.0 ** -52])
    pool = np.concatenate([special, rng.uniform(-3, 3, max(n_values - len(special), 0))])[:n_values]
    assert len(np.unique(pool.view(np.uint64))) == n_values
    val = pool[rng.integers(0, n_values, rp[-1])]
    val[:n_values] = pool  # every value occurs
    return CRS(n, rp, col, val)


@pytest.fixture
def no_sellwin(ctx):
    """The gather forms of the dictionary kernel (1-3) on their own: the x-window / sliced-ELL form (4-5,
    bis_spmv_sell.hip), which takes precedence where a matrix qualifies, is switched off for the test."""
    ctx.set_option("spmv_sellwin", 0)
    yield
    ctx.set_option("spmv_sellwin", -1)


@pytest.mark.parametrize("rp64", [0, 1])
@pytest.mark.parametrize("form", [1, 2])
def test_spmv_value_dictionary(ctx, oracle, form, rp64, no_sellwin):
    """Matrices with at most 256 distinct values (compared bit for bit) stream 1-byte value codes against a
    dictionary held in LDS (bis_mat_spmv_stream_info): a lossless re-encoding -- y is BIT-IDENTICAL to the kernel that
    streams the CRS values (same products, same summation order) and within the kernel tolerance of the oracle's
    kernels.hpp:22-52 loop.  Both forms of the kernel (1: consecutive non-zeros per lane, 2: a lane per row with the
    codes staged through LDS, the default where rows are short); 2, 256 and 257 distinct values; blocks of
    empty rows; 64-bit row pointers; a matrix whose values change in place (-scale) drops its dictionary; matrices whose
    OFF-DIAGONAL values are few while the diagonal is not (Anderson) keep a per-row diagonal array beside the
    dictionary, unless some row holds two diagonal entries."""
    rng = np.random.default_rng(40 + form)
    ctx.set_option("force_rp64", rp64)
    ctx.set_option("spmv_win8", 0)  # (the 8-byte baseline here is the row-block kernel on the CRS arrays; win8: tests/test_gpu_win8.py)
    try:
        dup = oracle.gen_anderson(9, W=5.0, shift=3.0)
        dup = CRS(dup.n_rows, dup.row_ptr, dup.col.copy(), dup.val)
        dup.col[dup.row_ptr[100]:dup.row_ptr[100] + 2] = 100  # row 100: two diagonal entries with different values
        # expected (val_bytes, dictionary values, kernel form) per option value; None: whatever the packing allows
        d = lambda n: {1: (1, n, 1), 2: (1, n, 2)}
        diag_only_rowmajor = lambda n: {1: (8, 0, 0), 2: (1, n, 3)}
        cases = [("hpcg", oracle.gen_hpcg(12, 10, 9), d(2)), ("anderson W=0", oracle.gen_anderson(9, W=0.0, shift=7.0), d(2)),
                 ("256 values", _few_values_matrix(rng, 9000, 256, empty_head=5000), d(256)),
                 ("257 values", _few_values_matrix(rng, 3000, 257), {1: (8, 0, 0), 2: (8, 0, 0)}),
                 # random diagonal, constant hopping: dictionary {-t} + per-row diagonal values (lane-per-row form only)
                 ("anderson W=5", oracle.gen_anderson(9, W=5.0, shift=3.0), diag_only_rowmajor(1)),
                 ("anderson 12, W=2", oracle.gen_anderson(12, t=0.5, W=2.0), diag_only_rowmajor(1)),
                 ("two diagonal entries in a row", dup, {1: (8, 0, 0), 2: (8, 0, 0)}),
                 ("fem", oracle.gen_fem(5, 4, 3), None)]
        for name, A, want in cases:
            x = rng.uniform(-1, 1, A.n_cols)
            ys = {}
            for mode in (0, form):
                ctx.set_option("spmv_valdict", mode)
                dA = ctx.matrix(A)
                col_b, val_b, n_dict, kform = dA.spmv_stream_info()
                if mode == 0:
                    assert (val_b, n_dict, kform) == (8, 0, 0)
                elif want is not None and col_b == 2:
                    assert (val_b, n_dict, kform) == want[form], name
                dx, dy = ctx.upload(x), ctx.alloc(A.n_rows)
                ctx.spmv(dA, dx, dy)
                ys[mode] = dy.to_host()
                dA.free(); dx.free(); dy.free()
            assert np.array_equal(ys[0], ys[form], equal_nan=True), name
            if "values" not in name and "entries" not in name:  # (the huge values of those overflow to inf/nan in both)
                yo = oracle.spmv(A, x)
                scale = np.abs(A.to_scipy()).dot(np.abs(x)).max()
                assert np.max(np.abs(ys[form] - yo)) <= KTOL * scale, name
        # in-place scaling after the dictionary was built
        ctx.set_option("spmv_valdict", form)
        A = oracle.gen_hpcg(8)
        dA = ctx.matrix(A)
        x = rng.uniform(-1, 1, A.n_rows)
        dx, dy = ctx.upload(x), ctx.alloc(A.n_rows)
        ctx.spmv(dA, dx, dy)
        assert dA.spmv_stream_info()[1:3] == (1, 2)
        s = ctx.scale_sym(dA)
        ctx.spmv(dA, dx, dy)
        sv = s.to_host()
        B = CRS(A.n_rows, A.row_ptr, A.col, A.val * sv[np.repeat(np.arange(A.n_rows), np.diff(A.row_ptr))] * sv[A.col])
        yo = oracle.spmv(B, x)
        assert np.max(np.abs(dy.to_host() - yo)) <= KTOL * np.abs(B.to_scipy()).dot(np.abs(x)).max()
    finally:
        ctx.set_option("spmv_valdict", -1)
        ctx.set_option("force_rp64", -1)
        ctx.set_option("spmv_win8", -1)


@pytest.mark.parametrize("form", [1, 2])
def test_value_dictionary_in_fused_cg_and_colour_sweeps(ctx, oracle, form, no_sellwin):
    """The dictionary kernel's other two epilogues: the fused (Ap, p) of the CG schedule (bis_cg.hip) and the
    triangular-sweep step on a colour block of a multi-coloured matrix.  The sweeps are bit-identical to the ones
    computed from the streamed CRS values; so is the CG history with form 1 (same row blocks, same partial sums of
    (Ap, p)); form 2 sums (Ap, p) over blocks of 256 rows, a different but equally fixed order: history within 1e-12 r0."""
    A = oracle.gen_hpcg(16)
    n = A.n_rows
    out = {}
    ctx.set_option("spmv_win8", 0)  # (the 8-byte baseline here is the row-block kernel on the CRS arrays)
    try:
        for mode in (0, form):
            ctx.set_option("spmv_valdict", mode)
            dA = ctx.gen_hpcg(16)
            b, x = ctx.alloc(n), ctx.alloc(n)
            ctx.init_vector(b, 1.0); ctx.init_vector(x, 0.1)
            cg = ctx.cg(dA, b, x)
            cg.init(0.0)
            cg.iterate(25)
            iters, conv, hist = cg.status(hist_cap=64)
            B, perm, n_col = ctx.multicolour(dA)
            Ls, Us, D, Dinv = ctx.split_strict(B)
            rhs = ctx.upload(np.random.default_rng(3).uniform(-1, 1, n))
            xf, xb = ctx.alloc(n), ctx.alloc(n)
            ctx.sptrsv(Ls, xf, D, rhs)
            ctx.bsptrsv(Us, xb, D, rhs)
            yB = ctx.alloc(n)
            ctx.spmv(B, rhs, yB)  # the permuted matrix: its rows reach into too many column runs for the 8-window packing
            out[mode] = (np.array(hist), x.to_host(), xf.to_host(), xb.to_host(), dA.spmv_stream_info(), yB.to_host(),
                         B.spmv_stream_info())
            cg.free()
        assert out[0][4][1:] == (8, 0, 0) and out[form][4][1:] == (1, 2, form)
        assert out[0][6][1:] == (8, 0, 0)
        if form == 2:  # the lane-per-row kernel brings its own packing (32 windows of 2048 columns where 8 x 8192 do not do)
            assert out[form][6] == (2, 1, 2, 2)
        assert np.array_equal(out[0][5], out[form][5])
        for k in range(4):
            if form == 2 and k < 2:
                tol = 1e-12 * out[0][0][0] if k == 0 else 1e-10 * np.max(np.abs(out[0][1]))
                assert np.max(np.abs(out[0][k] - out[form][k])) <= tol, k
            else:
                assert np.array_equal(out[0][k], out[form][k]), k
    finally:
        ctx.set_option("spmv_valdict", -1)
        ctx.set_option("spmv_win8", -1)


def test_value_dictionary_kernel_selection_and_shapes(ctx, oracle, no_sellwin):
    """Which dictionary kernel a matrix gets (bis_mat_spmv_stream_info) and that every choice computes the same y as
    the CRS-value kernel, bit for bit: rows longer than the lane-per-row form's 40 entries -> consecutive form;
    rectangular matrices (more columns than rows: a rank's [owned | halo] numbering) in both encodings; row counts
    that are not a multiple of the 256-row blocks; a single row; columns scattered over more windows than the packed
    stream has -> no packing, no dictionary."""
    rng = np.random.default_rng(77)

    def banded(n, n_cols, width, lens, values, diag_random=False):
        rp = np.concatenate([[0], np.cumsum(lens)])
        col = np.empty(rp[-1], dtype=np.int32)
        val = np.empty(rp[-1])
        for r in range(n):
            k0, k1 = rp[r], rp[r + 1]
            c = rng.choice(np.arange(max(0, r - width), min(n_cols, r + width + 1)), size=k1 - k0, replace=False)
            if diag_random and r not in c:
                c[0] = r  # every row has its diagonal entry
            c = np.sort(c)
            col[k0:k1] = c
            val[k0:k1] = rng.choice(values, size=k1 - k0)
            if diag_random:
                hit = np.nonzero(c == r)[0]
                val[k0 + hit] = rng.uniform(1, 2, len(hit))
        return CRS(n, rp, col, val, n_cols=n_cols)

    vals = np.array([-1.0, 0.5, 26.0])
    n = 1000
    cases = [
        ("rows of 41", banded(n, n, 60, np.full(n, 41), vals), (1, 3, 1)),
        ("rows of 40", banded(n, n, 60, np.full(n, 40), vals), (1, 3, 2)),
        ("rectangular", banded(n, n + 300, 30, rng.integers(1, 20, n), vals), (1, 3, 2)),
        ("rectangular, random diagonal", banded(n, n + 300, 30, rng.integers(3, 20, n), vals, diag_random=True), None),
        ("777 rows", banded(777, 777, 20, rng.integers(0, 15, 777), vals), (1, 3, 2)),
        ("one row", CRS(1, [0, 2], [0, 0], [2.0, 3.0]), (1, 2, 2)),
        ("scattered columns", CRS(600, np.arange(0, 601 * 9, 9), rng.integers(0, 200000, 5400).astype(np.int32),
                                  rng.choice(vals, 5400), n_cols=200000), None),
    ]
    try:
        for name, A, want in cases:
            x = rng.uniform(-1, 1, A.n_cols)
            ys = {}
            for mode in (0, -1):
                ctx.set_option("spmv_valdict", mode)
                dA = ctx.matrix(A)
                info = dA.spmv_stream_info()
                if mode == -1 and want is not None and info[0] == 2:
                    assert info[1:] == want, (name, info)
                if mode == -1 and name == "rectangular, random diagonal":
                    assert info[0] != 2 or info[1:] == (1, 3, 3), info
                if name == "scattered columns":
                    assert info == (4, 8, 0, 0), info
                dx, dy = ctx.upload(x), ctx.alloc(A.n_rows)
                ctx.spmv(dA, dx, dy)
                ys[mode] = dy.to_host()
                dA.free(); dx.free(); dy.free()
            assert np.array_equal(ys[0], ys[-1]), name
            yo = oracle.spmv(A, x)
            assert np.max(np.abs(ys[-1] - yo)) <= KTOL * max(np.abs(A.to_scipy()).dot(np.abs(x)).max(), 1e-300), name
    finally:
        ctx.set_option("spmv_valdict", -1)


def _banded_few_values(rng, n, n_values, offsets, max_len, ragged=4, empty_every=53, n_cols=None, diag_random=False, huge=True):
    """Rows of max_len - ragged .. max_len entries at columns r + (an offset of `offsets`), unsorted, duplicates
    allowed, some empty rows; values drawn from n_values bit patterns (with -0.0, a denormal, huge values)."""
    n_cols = n_cols or n
    lens = rng.integers(max(max_len - ragged, 0), max_len + 1, n)
    if empty_every:
        lens[::empty_every] = 0
    rp = np.concatenate([[0], np.cumsum(lens)])
    rows = np.repeat(np.arange(n), lens)
    col = np.clip(rows + rng.choice(offsets, rp[-1]), 0, n_cols - 1).astype(np.int32)
    special = np.array([-0.0, 0.0, 5e-324, -1.0, 26.0, 1.7976931348623157e308, -1.7976931348623157e308, 1.0 + 2.0 ** -52])
    if not huge:
        special = special[:5]
    pool = np.concatenate([special, rng.uniform(-3, 3, max(n_values - len(special), 0))])[:n_values]
    val = pool[rng.integers(0, n_values, rp[-1])]
    val[:n_values] = pool
    if diag_random:  # exactly one diagonal entry per non-empty row, with a value of its own
        for r in range(n):
            k0, k1 = rp[r], rp[r + 1]
            if k1 > k0:
                hit = col[k0:k1] == r
                col[k0:k1][hit] = min(r + 1, n_cols - 1) if r + 1 < n_cols else r - 1
                col[k0] = r
                val[k0] = rng.uniform(5, 6)
    return CRS(n, rp, col, val, n_cols=n_cols)


@pytest.mark.parametrize("rp64,rows,joint,pairs,masks", [(0, 1, -1, -1, -1), (1, 2, -1, -1, -1), (0, 4, -1, -1, -1), (1, 4, -1, -1, -1), (0, 2, -1, -1, 0), (1, 1, -1, -1, 0), (0, 2, -1, 0, -1),
                                                         (1, 1, -1, 0, -1), (0, 2, 0, 0, -1), (1, 1, 0, 0, -1)])
def test_spmv_sellwin_form_is_bit_identical(ctx, oracle, rp64, rows, joint, pairs, masks):
    """Forms 4 / 5 of the dictionary SpMV (bis_spmv_sell.hip): the block's x entries in an LDS window, the codes per
    64-row slice in lane order with neutral padding.  y is BIT-IDENTICAL to the kernel that streams the CRS values and
    within the kernel tolerance of the oracle (kernels.hpp:22-42): stencils, banded matrices with several column runs,
    ragged and empty rows, -0.0 / denormal / huge values and -0.0 row sums (the padding must not turn them into +0.0),
    dictionaries of <= 32 and of up to 255 values, the per-row diagonal form, odd sizes (a window granule that reaches
    past the last column), an x that is only 8-byte aligned, 64-bit row pointers; blocks of 256, of 512 and (the row-mask form
    only; other matrices then get 512) of 1024 rows (option spmv_sellwin_rows); the one-byte (column - row, value) pair codes of stencil matrices (at most 253 pairs,
    each running through every block's window in step with the rows; a pair with a gap in its run inside a block, or
    too many pairs, takes the next format: spmv_sellwin_pairs 0 forces that), in their place 32 bits per ROW -- which of the matrix'
    pairs the row has -- where the matrix has at most 32 pairs and the rows' columns ascend (spmv_sellwin_masks 0: never), the 16-bit joint codes (tables of at most
    8 entries) and the 12-byte chunks in their place (spmv_sellwin_joint 0); matrices that do not qualify (256 values: no free code for the padding; scattered
    columns; mostly-padding rows) keep the gather forms."""
    rng = np.random.default_rng(90 + rp64)
    ctx.set_option("force_rp64", rp64)
    ctx.set_option("spmv_sellwin_rows", rows)
    ctx.set_option("spmv_sellwin_joint", joint)
    ctx.set_option("spmv_sellwin_pairs", pairs)
    ctx.set_option("spmv_sellwin_masks", masks)
    offs_band = np.arange(-40, 41)
    offs_runs = np.concatenate([np.arange(-3, 4), np.arange(-3, 4) + 700, np.arange(-3, 4) - 700, np.arange(-3, 4) + 5000, np.arange(-3, 4) - 5000])
    neg0 = CRS(300, np.arange(0, 301 * 3, 3), np.repeat(np.arange(300), 3).astype(np.int32), np.tile([-0.0, 0.0, -0.0], 300))
    neg0.val[::3] = -1.0  # rows: -1*x + 0*x + -0*x
    # few pairs, but the pair (column - row = 1000) exists only for rows whose index mod 256 is < 100 or >= 200: its
    # columns leave a gap in the window of every block, slot != base + row
    ng = 6000
    has = (np.arange(ng) % 256 < 100) | (np.arange(ng) % 256 >= 200)
    g_len = 3 + has.astype(np.int64)
    g_rp = np.concatenate([[0], np.cumsum(g_len)])
    g_col = np.concatenate([np.clip(np.array([r - 1, r, r + 1] + ([r + 1000] if has[r] else [])), 0, ng - 1) for r in range(ng)]).astype(np.int32)
    gap = CRS(ng, g_rp, g_col, np.where(g_col == np.repeat(np.arange(ng), g_len), 4.0, -1.0))
    try:
        cases = [("hpcg 12x10x9", oracle.gen_hpcg(12, 10, 9), 4), ("hpcg 20", oracle.gen_hpcg(20), 4),
                 ("anderson W=0", oracle.gen_anderson(9, W=0.0, shift=7.0), 4),
                 ("anderson W=5", oracle.gen_anderson(9, W=5.0, shift=3.0), 5),
                 ("anderson 12 W=2", oracle.gen_anderson(12, t=0.5, W=2.0), 5),
                 ("band, 20 values", _banded_few_values(rng, 9001, 20, offs_band, 27), 4),
                 ("band, 33 values", _banded_few_values(rng, 5000, 33, offs_band, 12), 4),
                 ("band, 255 values", _banded_few_values(rng, 7000, 255, offs_band, 31, empty_every=0), 4),
                 ("band, 256 values", _banded_few_values(rng, 7000, 256, offs_band, 31, empty_every=0), 2),
                 ("five runs", _banded_few_values(rng, 20011, 9, offs_runs, 18, huge=False), 4),
                 ("five runs, random diagonal", _banded_few_values(rng, 12000, 9, offs_runs, 18, diag_random=True), None),
                 ("band, 40 values, random diagonal", _banded_few_values(rng, 3000, 40, offs_band, 10, diag_random=True), None),
                 ("rectangular", _banded_few_values(rng, 3000, 5, np.arange(0, 300), 9, n_cols=3300, huge=False), 4),
                 ("long rows", _banded_few_values(rng, 2000, 6, np.arange(-100, 101), 70, huge=False), 4),
                 ("mostly padding", _banded_few_values(rng, 40000, 6, offs_band, 30, ragged=30), 2),
                 ("scattered", _few_values_matrix(rng, 9000, 12), None),
                 ("-0.0 sums", neg0, 4), ("one row", CRS(1, [0, 2], [0, 0], [2.0, 3.0]), 4), ("gap in a pair's run", gap, 4)]
        for name, A, want in cases:
            x = rng.uniform(-1, 1, A.n_cols)
            if name == "-0.0 sums":
                x[:] = 0.0
                x[::2] = -0.0
            ys = {}
            for mode in (0, -1):
                ctx.set_option("spmv_valdict", mode)
                dA = ctx.matrix(A)
                info = dA.spmv_stream_info()
                if mode == -1 and want is not None:
                    assert info[3] == want, (name, info)
                    if want >= 4:  # streamed bytes per non-zero: 1 with the pair codes, 2 with the joint codes, 3 otherwise
                        stencil = name.startswith(("hpcg", "anderson", "-0.0", "one row"))
                        few = A.nnz > 0 and len(np.unique(A.val.view(np.uint64))) <= 6 and want == 4
                        if stencil and pairs != 0:
                            if masks != 0 and name.startswith(("hpcg", "anderson")) and want == 4:
                                assert info[:2] == (0, 0), (name, info)  # nothing per non-zero: 32 bits per row
                            else:
                                assert info[:2] in ((1, 0), (0, 0)) and (masks != 0 or info[0] == 1), (name, info)
                        elif (few and joint != 0) or (name.startswith("anderson") and joint != 0):
                            assert info[:2] == (2, 0), (name, info)
                        if name.startswith("gap"):
                            assert info[0] == 2, (name, info)
                if mode == -1 and "random diagonal" in name:
                    assert info[3] in (5, 3, 0), (name, info)
                dx, dy = ctx.upload(x), ctx.alloc(A.n_rows)
                ctx.spmv(dA, dx, dy)
                ys[mode] = dy.to_host()
                if mode == -1:  # an x that is only 8-byte aligned: the window is filled through registers
                    dx1 = ctx.upload(np.concatenate([[7.0], x]))
                    ctx.init_vector(dy, 3.0)
                    ctx.spmv(dA, dx1.offset(1), dy)
                    ys["unaligned"] = dy.to_host()
                    dx1.free()
                dA.free(); dx.free(); dy.free()
            assert np.array_equal(ys[0].view(np.uint64), ys[-1].view(np.uint64)) or \
                (np.array_equal(ys[0], ys[-1], equal_nan=True) and np.array_equal(np.signbit(ys[0]), np.signbit(ys[-1]))), name
            assert np.array_equal(ys[-1], ys["unaligned"], equal_nan=True), name
            if np.all(np.abs(A.val) < 1e6):  # (the huge values of the other cases overflow to inf / nan in both)
                yo = oracle.spmv(A, x)
                scale = max(np.abs(A.to_scipy()).dot(np.abs(x)).max(), 1e-300)
                assert np.max(np.abs(ys[-1] - yo)) <= KTOL * scale, name
    finally:
        ctx.set_option("spmv_valdict", -1)
        ctx.set_option("force_rp64", -1)
        ctx.set_option("spmv_sellwin_rows", -1)
        ctx.set_option("spmv_sellwin_joint", -1)
        ctx.set_option("spmv_sellwin_pairs", -1)
        ctx.set_option("spmv_sellwin_masks", -1)


def _pair_matrix(rng, n, offsets, values, p_keep, sort=True, dup=False, empty_every=37):
    """Row r has the entries {(r + offsets[e], values[e]) : e kept with probability p_keep, column inside the matrix} -- at most
    len(offsets) distinct (column - row, value) pairs in the whole matrix; ascending columns unless sort is False; dup: some rows
    hold one of their entries twice."""
    offsets, values = np.asarray(offsets), np.asarray(values, dtype=np.float64)
    order = np.argsort(offsets, kind="stable")
    offsets, values = offsets[order], values[order]
    cols, vals, lens = [], [], []
    for r in range(n):
        keep = (rng.uniform(size=len(offsets)) < p_keep) & (r + offsets >= 0) & (r + offsets < n)
        if empty_every and r % empty_every == 5:
            keep[:] = False
        c, v = (r + offsets[keep]).astype(np.int32), values[keep]
        if dup and len(c) and r % 11 == 3:
            c, v = np.concatenate([c, c[-1:]]), np.concatenate([v, v[-1:]])
        if not sort and len(c) > 1 and r % 7 == 2:
            c, v = c[::-1], v[::-1]
        cols.append(c); vals.append(v); lens.append(len(c))
    rp = np.concatenate([[0], np.cumsum(lens)])
    return CRS(n, rp, np.concatenate(cols).astype(np.int32), np.concatenate(vals))


@pytest.mark.parametrize("rp64,rows", [(0, 1), (1, 2), (0, 4)])
def test_spmv_row_mask_form_randomised(ctx, oracle, rp64, rows):
    """The row-mask form (bis_spmv_sell.hip, spmv_sellmask_kernel: 32 bits per row over the matrix' <= 32 (column - row, value)
    pairs) on matrices that are NOT stencils of a grid: rows with random subsets of the pairs, empty rows, 1 .. 32 pairs (pair
    counts that are no multiple of the four pairs a step takes), two pairs with the same offset and different values, values
    -0.0 / denormal / huge / infinite; x with infinities and NaNs -- an entry a row does not have must not touch its sum
    (0 * inf), an entry it has must, exactly as in the CRS kernel.  y bit-identical to the kernel that streams the CRS
    values.  33 pairs, rows with descending columns, a column twice in a row: the form must not be chosen (the byte codes or
    the gather kernels run, same y)."""
    rng = np.random.default_rng(700 + rp64 + rows)
    ctx.set_option("force_rp64", rp64)
    ctx.set_option("spmv_sellwin_rows", rows)
    try:
        offs32 = np.concatenate([np.arange(-8, 8), 300 + np.arange(8), -300 - np.arange(8)])
        vals32 = np.array([26.0, -1.0, -0.0, 5e-324, 1.7976931348623157e308, -2.5, np.inf, 3.0])[rng.integers(0, 8, 32)]
        cases = [("1 pair", _pair_matrix(rng, 3000, [0], [2.0], 1.0), True),
                 ("3 pairs", _pair_matrix(rng, 5000, [-1, 0, 1], [-1.0, 4.0, -1.0], 0.9), True),
                 ("6 pairs, two per offset", _pair_matrix(rng, 4000, [-2, -2, 0, 0, 5, 5], [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], 0.5), True),
                 ("27 pairs", _pair_matrix(rng, 9001, offs32[:27], vals32[:27], 0.8), True),
                 ("32 pairs", _pair_matrix(rng, 20011, offs32, vals32, 0.7), True),
                 ("32 pairs, sparse rows", _pair_matrix(rng, 6000, offs32, vals32, 0.15), None),  # (mostly padding: any form)
                 ("33 pairs", _pair_matrix(rng, 6000, np.concatenate([offs32, [600]]), np.concatenate([vals32, [7.0]]), 0.8), False),
                 ("descending rows", _pair_matrix(rng, 6000, offs32[:9], vals32[:9], 0.9, sort=False), False),
                 ("a column twice", _pair_matrix(rng, 6000, offs32[:9], vals32[:9], 0.9, dup=True), False)]
        for name, A, want_masks in cases:
            x = rng.uniform(-1, 1, A.n_cols)
            x[rng.integers(0, A.n_cols, 40)] = np.inf
            x[rng.integers(0, A.n_cols, 40)] = np.nan
            x[rng.integers(0, A.n_cols, 40)] = -0.0
            ys = {}
            for mode in (0, -1):
                ctx.set_option("spmv_valdict", mode)
                dA = ctx.matrix(A)
                dx, dy = ctx.upload(x), ctx.alloc(A.n_rows)
                ctx.init_vector(dy, 7.0)
                ctx.spmv(dA, dx, dy)
                ys[mode] = dy.to_host()
                if mode == -1:
                    info = dA.spmv_stream_info()
                    if want_masks is True:
                        assert info[:2] == (0, 0) and info[3] == 4, (name, info)
                    elif want_masks is False:
                        assert info[:2] != (0, 0), (name, info)
                dA.free(); dx.free(); dy.free()
            # (bit for bit where the sum is a number or an infinity; a NaN where the CRS kernel has a NaN -- which of two NaN operands
            # an addition hands on, and with it the NaN's sign, depends on the operand order of the instruction, not on the arithmetic)
            nan0, nan1 = np.isnan(ys[0]), np.isnan(ys[-1])
            assert np.array_equal(nan0, nan1), name
            assert np.array_equal(ys[0].view(np.uint64)[~nan0], ys[-1].view(np.uint64)[~nan0]), name
            assert nan0.any() and np.isinf(ys[0]).any(), name  # (the case does exercise them)
    finally:
        ctx.set_option("spmv_valdict", -1)
        ctx.set_option("force_rp64", -1)
        ctx.set_option("spmv_sellwin_rows", -1)


@pytest.mark.parametrize("rows,masks", [(1, -1), (2, -1), (4, -1), (2, 0)])
def test_spmv_sellwin_in_fused_cg(ctx, oracle, rows, masks):
    """The fused (Ap, p) epilogue of form 4 sums over the same 256-row blocks and waves as the lane-per-row gather
    form: the CG history is bit-identical to it, and within 1e-10 r0 of the oracle's (methods/cg.hpp:6-54);
    in-place scaling drops the form and the next SpMV rebuilds it from the new values."""
    n1 = 24
    A = oracle.gen_hpcg(n1)
    n = A.n_rows
    hists = {}
    ctx.set_option("spmv_sellwin_rows", rows)
    ctx.set_option("spmv_sellwin_masks", masks)
    try:
        for sw in (0, -1):
            ctx.set_option("spmv_sellwin", sw)
            dA = ctx.gen_hpcg(n1)
            assert dA.spmv_stream_info()[3] == (2 if sw == 0 else 4)
            b, x = ctx.alloc(n), ctx.alloc(n)
            ctx.init_vector(b, 1.0); ctx.init_vector(x, 0.1)
            cg = ctx.cg(dA, b, x)
            cg.init(1e-14)
            cg.iterate(80)
            iters, conv, hist = cg.status(hist_cap=128)
            hists[sw] = (iters, np.array(hist), x.to_host())
            cg.free(); dA.free(); b.free(); x.free()
        assert hists[0][0] == hists[-1][0] and np.array_equal(hists[0][1], hists[-1][1]) and np.array_equal(hists[0][2], hists[-1][2])
        ref = oracle.solve(A, "cg", "none")
        m = min(len(ref["hist"]), len(hists[-1][1]))
        assert np.max(np.abs(ref["hist"][:m] - hists[-1][1][:m])) <= 1e-10 * ref["hist"][0]
        ctx.set_option("spmv_sellwin", -1)
        A8 = oracle.gen_hpcg(8)
        dA = ctx.matrix(A8)
        xh = np.random.default_rng(5).uniform(-1, 1, A8.n_rows)
        dx, dy = ctx.upload(xh), ctx.alloc(A8.n_rows)
        ctx.spmv(dA, dx, dy)
        assert dA.spmv_stream_info()[3] == 4
        sv = ctx.scale_sym(dA).to_host()
        ctx.spmv(dA, dx, dy)
        B = CRS(A8.n_rows, A8.row_ptr, A8.col, A8.val * sv[np.repeat(np.arange(A8.n_rows), np.diff(A8.row_ptr))] * sv[A8.col])
        assert np.max(np.abs(dy.to_host() - oracle.spmv(B, xh))) <= KTOL * np.abs(B.to_scipy()).dot(np.abs(xh)).max()
    finally:
        ctx.set_option("spmv_sellwin", -1)
        ctx.set_option("spmv_sellwin_rows", -1)
        ctx.set_option("spmv_sellwin_masks", -1)


@pytest.mark.parametrize("n", [1, 2, 777, 4096, 100001, 1 << 20])
def test_fused_axpy_dot_is_bit_identical_to_the_separate_kernels(ctx, n):
    """bis_axpy_dot_dev (the modified Gram-Schmidt axpy of step j fused with the dot of step j+1, gmres.hpp:13-14,:25,
    and with the sum of squares, :36-38): same w and the same reduction value, bit for bit, as
    bis_subtract_vectors_dev followed by bis_dot_dev -- for even and odd n and for basis vectors at odd offsets
    (V + j n with odd n: every other vector is only 8-byte aligned)."""
    import ctypes as C
    lib = ctx.lib
    rng = np.random.default_rng(n)
    V = ctx.upload(rng.uniform(-1, 1, 3 * n + 1))
    w0 = rng.uniform(-1, 1, n)
    sc = ctx.upload(np.array([0.37, 0.0, 0.0]))
    p = lambda vec, off=0: C.c_void_p(vec.ptr + 8 * off)
    for u_off, v_off in ((0, n), (n, 2 * n), (1, n + 1), (0, None)):
        wa, wb = ctx.upload(w0), ctx.upload(w0)
        vp = p(V, v_off) if v_off is not None else None
        ctx.check(lib.bis_subtract_vectors_dev(ctx.h, p(wa), p(wa), p(V, u_off), C.c_int64(n), p(sc)))
        ctx.check(lib.bis_dot_dev(ctx.h, p(wa), vp if vp is not None else p(wa), C.c_int64(n), p(sc, 1)))
        ctx.check(lib.bis_axpy_dot_dev(ctx.h, p(wb), p(V, u_off), p(sc), vp, C.c_int64(n), p(sc, 2)))
        ctx.sync()
        s = sc.to_host()
        assert np.array_equal(wa.to_host(), wb.to_host())
        assert s[1] == s[2] and np.isfinite(s[1])
        wa.free(); wb.free()


@pytest.mark.parametrize("seed", range(12))
def test_spmv_formats_randomised_differential(ctx, oracle, seed):
    """Seeded random matrices across the format decisions (row lengths around the lane-per-row limit, value pools
    around 255 / 256 / 257, banded / windowed / scattered columns, with and without a full diagonal, empty rows, 32-
    and 64-bit row pointers, rectangular): whatever kernel the library picks, y equals the CRS-value kernel's bit for
    bit and the oracle's within the kernel tolerance."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 6000))
    n_cols = n + int(rng.integers(0, 2)) * int(rng.integers(0, 500))
    max_len = int(rng.choice([1, 3, 7, 27, 39, 40, 41, 45]))
    lens = rng.integers(0, max_len + 1, n)
    if rng.integers(0, 2):
        lens[rng.integers(0, n, max(1, n // 10))] = 0
    lens = np.minimum(lens, n_cols)
    pool_n = int(rng.choice([1, 2, 17, 255, 256, 257, 400]))
    pool = np.unique(rng.uniform(-4, 4, pool_n + 8))[:pool_n]
    style = int(rng.integers(0, 3))  # 0 band around the row, 1 a few far windows, 2 scattered
    full_diag = bool(rng.integers(0, 2)) and n_cols >= n
    rp = np.concatenate([[0], np.cumsum(lens)])
    col = np.empty(rp[-1], dtype=np.int32)
    val = np.empty(rp[-1])
    offs = rng.integers(0, max(n_cols - 64, 1), 5)
    for r in range(n):
        k0, k1 = rp[r], rp[r + 1]
        m = k1 - k0
        if m == 0:
            continue
        if style == 0:
            cand = np.arange(max(0, r - 60), min(n_cols, r + 61))
        elif style == 1:
            cand = np.unique(np.concatenate([(o + np.arange(64)) % n_cols for o in offs] + [np.array([min(r, n_cols - 1)])]))
        else:
            cand = np.arange(n_cols)
        c = rng.choice(cand, size=min(m, len(cand)), replace=False)
        if len(c) < m:
            c = np.concatenate([c, rng.choice(cand, size=m - len(c))])  # duplicates allowed
        if full_diag and r not in c:
            c[0] = r
        col[k0:k1] = np.sort(c) if rng.integers(0, 2) else c
        val[k0:k1] = rng.choice(pool, size=m)
        if full_diag:
            hit = np.nonzero(col[k0:k1] == r)[0]
            val[k0 + hit[:1]] = rng.uniform(5, 6)  # a distinct diagonal value per row
    A = CRS(n, rp, col, val, n_cols=n_cols)
    x = rng.uniform(-1, 1, n_cols)
    ctx.set_option("force_rp64", int(rng.integers(0, 2)))
    try:
        ys = {}
        for mode in (0, -1):
            ctx.set_option("spmv_valdict", mode)
            dA = ctx.matrix(A)
            info = dA.spmv_stream_info()
            dx, dy = ctx.upload(x), ctx.alloc(n)
            ctx.spmv(dA, dx, dy)
            ctx.spmv(dA, dx, dy)  # the second call reuses the structures the first one built
            ys[mode] = dy.to_host()
            dA.free(); dx.free(); dy.free()
        assert np.array_equal(ys[0], ys[-1]), (seed, info, n, max_len, pool_n, style, full_diag)
        yo = oracle.spmv(A, x)
        assert np.max(np.abs(ys[-1] - yo)) <= KTOL * max(np.abs(A.to_scipy()).dot(np.abs(x)).max(), 1e-300)
    finally:
        ctx.set_option("spmv_valdict", -1)
        ctx.set_option("force_rp64", -1)


@pytest.mark.parametrize("seed", range(10))
def test_tiled_sweep_random_stencils(ctx, oracle, capfd, monkeypatch, seed):
    """The device-built tile plan on random stencils: a random grid (extents, unknowns per node), a random subset of
    the neighbour offsets within distance 2 that precede a row in the natural order (some need skews, some admit none
    within the search range), random values, as an uploaded matrix with a grid hint.  Forward sweep of the lower and
    backward sweep of the transposed (upper) pattern: where the plan applies bit-exact against the fma oracle, where it
    is refused the level kernels' result within their tolerance."""
    monkeypatch.setenv("BIS_TRSV_TILE_STATS", "1")
    rng = np.random.default_rng(500 + seed)
    nx, ny, nz = (int(rng.integers(3, 14)) for _ in range(3))
    dof = int(rng.choice([1, 1, 2, 3]))
    reach = int(rng.choice([1, 1, 2]))
    cand = [(dx, dy, dz, dd) for dz in range(-reach, reach + 1) for dy in range(-reach, reach + 1)
            for dx in range(-reach, reach + 1) for dd in range(-(dof - 1), dof)
            if (dz, dy, dx, dd) < (0, 0, 0, 0)]
    pick = rng.random(len(cand)) < rng.uniform(0.15, 0.7)
    sten = [c for c, p in zip(cand, pick) if p] or [cand[-1]]
    n = nx * ny * nz * dof
    rows = [[] for _ in range(n)]
    for z in range(nz):
        for y in range(ny):
            for x in range(nx):
                for d in range(dof):
                    r = ((z * ny + y) * nx + x) * dof + d
                    for dx, dy, dz, dd in sten:
                        X, Y, Z, Dd = x + dx, y + dy, z + dz, d + dd
                        if 0 <= X < nx and 0 <= Y < ny and 0 <= Z < nz and 0 <= Dd < dof:
                            rows[r].append(((Z * ny + Y) * nx + X) * dof + Dd)
    for r in range(n):
        rows[r].sort()
    rp = np.concatenate([[0], np.cumsum([len(c) for c in rows])])
    col = np.array([c for cs in rows for c in cs], dtype=np.int32)
    L = CRS(n, rp, col, rng.uniform(-0.3, 0.3, len(col)))
    U = L.to_scipy().T.tocsr()
    U.sort_indices()
    U = CRS(n, U.indptr, U.indices.astype(np.int32), U.data)
    D, b = rng.uniform(1, 2, n), rng.uniform(-1, 1, n)
    want_f, want_b = oracle.sptrsv(L, D, b), oracle.sptrsv(U, D, b, backward=True)
    dD, db, x = ctx.upload(D), ctx.upload(b), ctx.alloc(n)
    got, planned = {}, {}
    try:
        for mode in (0, -1):
            ctx.set_option("trsv_tiled", mode)
            dL, dU = ctx.matrix(L), ctx.matrix(U)
            dL.set_grid_hint(nx, ny, nz, dof); dU.set_grid_hint(nx, ny, nz, dof)
            capfd.readouterr()
            ctx.sptrsv(dL, x, dD, db)
            f = x.to_host()
            ctx.bsptrsv(dU, x, dD, db)
            got[mode] = (f, x.to_host())
            planned[mode] = capfd.readouterr().err.count("tiled sptrsv plan")
            dL.free(); dU.free()
        assert planned[0] == 0 and planned[-1] in (0, 2)
        scale = max(np.max(np.abs(want_f)), np.max(np.abs(want_b)))
        for mode in (0, -1):
            for k, want in enumerate((want_f, want_b)):
                if planned[mode]:  # the tiled sweep keeps the reference's CRS-order fma chain: bit-exact
                    assert np.array_equal(got[mode][k], want), (seed, mode, k, nx, ny, nz, dof, sten)
                else:  # small grids fall to the few-level kernels (products, then sums): kernel tolerance
                    assert np.max(np.abs(got[mode][k] - want)) <= 1e-12 * scale, (seed, mode, k)
    finally:
        ctx.set_option("trsv_tiled", -1)
    capfd.readouterr()
