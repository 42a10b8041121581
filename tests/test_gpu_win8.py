"""GPU: the window + sliced-ELL SpMV with the 8-byte values streamed ("win8", bis_spmv_sell.hip; round 5): the kernel that
serves matrices with ARBITRARY values (no dictionary) whose blocks read a few contiguous runs of x.  y is BIT-IDENTICAL to
the row-block kernel on the CRS arrays (spmv_win8 = 0) and within 1e-13 of the oracle (kernels.hpp:22-42): stencils with
random values, banded matrices with several column runs, ragged / empty rows, unsorted columns and duplicates, -0.0 row
sums, odd sizes, an x that is only 8-byte aligned, 64-bit row pointers, blocks of 256 / 512 / 1024 rows, rings of 2..6
chunks; the fused (Ap, p) epilogue inside the CG loop; matrices whose plan does not apply keep the gather kernel."""
import numpy as np
import pytest

from oracle.pyoracle import CRS

pytestmark = pytest.mark.gpu
KTOL = 1e-13


@pytest.fixture(scope="module")
def ctx():
    from basic_iterative_solvers_amd import Context
    c = Context()
    yield c
    c.close()


def banded(rng, n, offsets, max_len, ragged=4, empty_every=53, n_cols=None, special=True):
    n_cols = n_cols or n
    lens = rng.integers(max(max_len - ragged, 0), max_len + 1, n)
    if empty_every:
        lens[::empty_every] = 0
    rp = np.concatenate([[0], np.cumsum(lens)])
    rows = np.repeat(np.arange(n), lens)
    col = np.clip(rows + rng.choice(offsets, rp[-1]), 0, n_cols - 1).astype(np.int32)
    val = rng.uniform(-3, 3, rp[-1])
    if special and rp[-1] > 8:
        val[:8] = [-0.0, 0.0, 5e-324, -1.0, 26.0, 1.7976931348623157e308, -1.7976931348623157e308, 1.0 + 2.0 ** -52]
    return CRS(n, rp, col, val, n_cols=n_cols)


def _two_populations(rng, n=30000):
    """every 64th row long, the others short: whatever the order inside a block, the plan must stay correct"""
    lens = rng.integers(0, 9, n)
    lens[::64] = 200
    rp = np.concatenate([[0], np.cumsum(lens)])
    rows = np.repeat(np.arange(n), lens)
    col = np.clip(rows + rng.integers(-300, 301, rp[-1]), 0, n - 1).astype(np.int32)
    return n, rp, col, rng.uniform(-1, 1, rp[-1])


def randomised(A, rng):
    """the same pattern with random values (a stencil generator's matrix without its few values)"""
    return CRS(A.n_rows, A.row_ptr, A.col, rng.uniform(-2, 2, A.nnz), n_cols=A.n_cols)


@pytest.mark.parametrize("rp64,rows,depth", [(0, 1, 4), (0, 2, 4), (1, 2, 2), (0, 4, 3), (1, 1, 6), (0, 2, 6)])
def test_win8_is_bit_identical_to_the_rowblock_kernel(ctx, oracle, rp64, rows, depth):
    rng = np.random.default_rng(500 + rows + 10 * rp64)
    offs_band = np.arange(-40, 41)
    offs_runs = np.concatenate([np.arange(-3, 4), np.arange(-3, 4) + 700, np.arange(-3, 4) - 700, np.arange(-3, 4) + 5000, np.arange(-3, 4) - 5000])
    neg0 = CRS(300, np.arange(0, 301 * 3, 3), np.repeat(np.arange(300), 3).astype(np.int32), np.tile([-1.0, 0.0, -0.0], 300))
    cases = [("hpcg 12x10x9, random values", randomised(oracle.gen_hpcg(12, 10, 9), rng), 6),
             ("hpcg 20", oracle.gen_hpcg(20), 6),
             ("anderson 12 raw (7 entries in 8 slots)", oracle.gen_anderson(12), None),
             ("fem 6x5x4", oracle.gen_fem(6, 5, 4), None),
             ("fem 10x9x8", oracle.gen_fem(10, 9, 8), None),
             ("band", banded(rng, 9001, offs_band, 27), 6),
             ("band, no empty rows", banded(rng, 7000, offs_band, 31, empty_every=0), 6),
             ("five runs", banded(rng, 20011, offs_runs, 18, special=False), 6),
             ("rectangular", banded(rng, 3000, np.arange(0, 300), 9, n_cols=3300, special=False), 6),
             ("long rows", banded(rng, 2000, np.arange(-100, 101), 70, special=False), 6),
             ("ragged rows (sorted by length inside a block)", banded(rng, 40000, offs_band, 30, ragged=30), None),
             ("rows of 0..8 entries among rows of 200", CRS(*_two_populations(rng)), None),
             ("scattered", CRS(9000, np.arange(0, 9001 * 12, 12), rng.integers(0, 9000, 9000 * 12).astype(np.int32), rng.uniform(-1, 1, 9000 * 12)), 0),
             ("-0.0 sums", neg0, 6), ("one row", CRS(1, [0, 2], [0, 0], [2.0, 3.0]), 6)]
    ctx.set_option("force_rp64", rp64)
    ctx.set_option("spmv_valdict", 0)  # no dictionary forms: win8 or the row-block kernel on the CRS arrays
    ctx.set_option("spmv_win8_rows", rows)
    ctx.set_option("spmv_win8_depth", depth)
    try:
        for name, A, want in cases:
            x = rng.uniform(-1, 1, A.n_cols)
            if name == "-0.0 sums":
                x[:] = 0.0
                x[::2] = -0.0
            ys = {}
            for w8 in (0, -1):
                ctx.set_option("spmv_win8", w8)
                dA = ctx.matrix(A)
                info = dA.spmv_stream_info()
                assert info[1] == 8, (name, info)
                if w8 == 0:
                    assert info[3] == 0, (name, info)
                elif want is not None:
                    assert info[3] == want, (name, info)
                dx, dy = ctx.upload(x), ctx.alloc(A.n_rows)
                ctx.init_vector(dy, float("nan"))
                ctx.spmv(dA, dx, dy)
                ys[w8] = dy.to_host()
                if w8 == -1:  # an x that is only 8-byte aligned: the window is filled through registers
                    dx1 = ctx.upload(np.concatenate([[7.0], x]))
                    ctx.init_vector(dy, 3.0)
                    ctx.spmv(dA, dx1.offset(1), dy)
                    ys["unaligned"] = dy.to_host()
                    dx1.free()
                    ctx.spmv(dA, dx, dy)  # a second product on the built form
                    assert np.array_equal(dy.to_host(), ys[-1], equal_nan=True), name
                dA.free(); dx.free(); dy.free()
            same_bits = np.array_equal(ys[0].view(np.uint64), ys[-1].view(np.uint64))
            assert same_bits or (np.array_equal(ys[0], ys[-1], equal_nan=True) and np.array_equal(np.signbit(ys[0]), np.signbit(ys[-1]))), name
            assert np.array_equal(ys[-1], ys["unaligned"], equal_nan=True), name
            if np.all(np.abs(A.val) < 1e6):
                yo = oracle.spmv(A, x)
                scale = max(np.abs(A.to_scipy()).dot(np.abs(x)).max(), 1e-300)
                assert np.max(np.abs(ys[-1] - yo)) <= KTOL * scale, name
    finally:
        for k in ("force_rp64", "spmv_valdict", "spmv_win8_rows", "spmv_win8_depth", "spmv_win8"):
            ctx.set_option(k, -1)


def _hpcg_random_diagonal(oracle, n1, seed):
    """the HPCG operator plus a random non-negative diagonal: SPD, arbitrary values (no dictionary form), 27 entries per row"""
    A = oracle.gen_hpcg(n1)
    rows = np.repeat(np.arange(A.n_rows), np.diff(A.row_ptr))
    val = A.val.copy()
    val[A.col == rows] += np.random.default_rng(seed).uniform(0, 1, A.n_rows)
    return CRS(A.n_rows, A.row_ptr, A.col, val)


@pytest.mark.parametrize("rows", [1, 2, 4])
def test_win8_in_fused_cg(ctx, oracle, rows):
    """The fused (Ap, p) epilogue of win8 inside the device CG schedule (methods/cg.hpp:6-54) on an SPD matrix with arbitrary
    values: within 1e-10 r0 of the oracle's history, like the run on the row-block kernel; in-place scaling drops the form and
    the next SpMV rebuilds it from the new values."""
    A = _hpcg_random_diagonal(oracle, 20, 3)
    n = A.n_rows
    hists = {}
    ctx.set_option("spmv_valdict", 0)
    ctx.set_option("spmv_win8_rows", rows)
    try:
        for w8 in (0, -1):
            ctx.set_option("spmv_win8", w8)
            dA = ctx.matrix(A)
            assert dA.spmv_stream_info()[3] == (0 if w8 == 0 else 6)
            b, x = ctx.alloc(n), ctx.alloc(n)
            ctx.init_vector(b, 1.0); ctx.init_vector(x, 0.1)
            cg = ctx.cg(dA, b, x)
            cg.init(1e-14)
            cg.iterate(80)
            iters, conv, hist = cg.status(hist_cap=128)
            hists[w8] = (iters, np.array(hist), x.to_host())
            cg.free(); dA.free(); b.free(); x.free()
        ref = oracle.solve(A, "cg", "none")
        for w8 in (0, -1):
            m = min(len(ref["hist"]), len(hists[w8][1]))
            assert np.max(np.abs(ref["hist"][:m] - hists[w8][1][:m])) <= 1e-10 * ref["hist"][0]
            assert abs(hists[w8][0] - ref["iters"]) <= 1
        ctx.set_option("spmv_win8", -1)
        A8 = _hpcg_random_diagonal(oracle, 8, 4)
        dA = ctx.matrix(A8)
        xh = np.random.default_rng(5).uniform(-1, 1, A8.n_rows)
        dx, dy = ctx.upload(xh), ctx.alloc(A8.n_rows)
        ctx.spmv(dA, dx, dy)
        assert dA.spmv_stream_info()[3] == 6
        sv = ctx.scale_sym(dA).to_host()
        ctx.spmv(dA, dx, dy)
        B = CRS(A8.n_rows, A8.row_ptr, A8.col, A8.val * sv[np.repeat(np.arange(A8.n_rows), np.diff(A8.row_ptr))] * sv[A8.col])
        assert np.max(np.abs(dy.to_host() - oracle.spmv(B, xh))) <= KTOL * np.abs(B.to_scipy()).dot(np.abs(xh)).max()
    finally:
        for k in ("spmv_valdict", "spmv_win8_rows", "spmv_win8"):
            ctx.set_option(k, -1)


def test_win8_placement_search_and_its_record(ctx, oracle):
    """A stream of >= 1 GiB is placement-searched when it is built (bis_spmv_sell.hip w8_tune_placement): the search keeps an
    allocation at least as fast as the first one, reports what it did (bis_mat_win8_tuning), can be switched off, and whichever
    allocation it keeps holds the same stream: y is bit-identical.  (HPCG-160: 4.1 M rows, 1.14 GiB of stream.)"""
    n1 = 160
    N = n1 ** 3
    x = ctx.upload(np.random.default_rng(8).uniform(-1, 1, N))
    ys = {}
    ctx.set_option("spmv_valdict", 0)
    try:
        for tune in (0, 4, -1):
            ctx.set_option("spmv_win8_tune", tune)
            dA = ctx.gen_hpcg(n1)
            y = ctx.alloc(N)
            ctx.spmv(dA, x, y)
            assert dA.spmv_stream_info()[3] == 6
            trials, first_ms, kept_ms = dA.win8_tuning()
            if tune == 0:
                assert (trials, first_ms, kept_ms) == (0, 0.0, 0.0)
            else:
                assert 0 <= trials <= (4 if tune == 4 else 12) and first_ms > 0 and 0 < kept_ms <= first_ms
            ys[tune] = y.to_host()
            dA.free(); y.free()
        assert np.array_equal(ys[0], ys[4]) and np.array_equal(ys[0], ys[-1])
        A = oracle.gen_hpcg(n1, row0=1000000, row1=1050000)
        yo = oracle.spmv(A, x.to_host())
        assert np.max(np.abs(ys[-1][1000000:1050000] - yo)) <= KTOL * np.max(np.abs(yo))
    finally:
        ctx.set_option("spmv_valdict", -1)
        ctx.set_option("spmv_win8_tune", -1)
    x.free()
