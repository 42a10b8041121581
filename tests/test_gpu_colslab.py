"""GPU: column slabs (bis_spmv_slab.hip; round 5) -- the SpMV's form for a matrix WITHOUT locality.  The matrix is cut into
K column ranges whose x slices fit an XCD's L2 and multiplied in K passes, each continuing the rows' left-to-right sums
(kernels.hpp:25-39) where the pass before left them: y is BIT-IDENTICAL to the one-pass row-block kernel and within 1e-13 of
the oracle.  Forced slab counts on small matrices (ragged and empty rows, empty slabs, -0.0 sums, rectangular shapes, more
slabs than columns would need), rows that are not ascending (plan refused), the fused (Ap, p) epilogue of the CG loop in
the last pass, values scaled in place (slabs rebuilt), and config 5's unstructured input at full size, where the build-time
trial picks the slabs by itself."""
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


def scattered(rng, n, max_len, n_cols=None, empty_every=41, col_hi=None, sort=True):
    """rows of 0..max_len entries at random columns (no locality), ascending inside a row unless sort=False"""
    n_cols = n_cols or n
    lens = rng.integers(0, max_len + 1, n)
    if empty_every:
        lens[::empty_every] = 0
    cols = []
    for ln in lens:
        c = np.unique(rng.integers(0, col_hi or n_cols, int(ln)))  # (ascending, distinct)
        cols.append(c if sort else rng.permutation(c))
    rp = np.concatenate([[0], np.cumsum([len(c) for c in cols])])
    col = (np.concatenate(cols) if rp[-1] else np.zeros(0)).astype(np.int32)
    val = rng.uniform(-3, 3, rp[-1])
    if rp[-1] > 8:
        val[:8] = [-0.0, 0.0, 5e-324, -1.0, 26.0, 1.7976931348623157e308, -1.7976931348623157e308, 1.0 + 2.0 ** -52]
    return CRS(n, rp, col, val, n_cols=n_cols)


def _y(ctx, A, xh, want_form=None, want_k=None):
    dA = ctx.matrix(A)
    dx, dy = ctx.upload(xh), ctx.alloc(A.n_rows)
    ctx.init_vector(dy, 7.0)  # (a pass that forgot to start from zero would show)
    ctx.spmv(dA, dx, dy)
    ctx.spmv(dA, dx, dy)
    y = dy.to_host()
    form = dA.spmv_stream_info()[3]
    k = dA.colslab_info()[0]
    dA.free(); dx.free(); dy.free()
    if want_form is not None:
        assert form == want_form, (form, want_form)
    if want_k is not None:
        assert k == want_k
    return y


@pytest.mark.parametrize("K", [2, 3, 7, 32])
def test_colslab_bit_identical_to_one_pass(ctx, oracle, K):
    rng = np.random.default_rng(40 + K)
    neg0 = CRS(300, np.arange(0, 301 * 3, 3), np.tile([5, 150, 290], 300).astype(np.int32), np.tile([-1.0, 0.0, -0.0], 300))
    cases = [("square, ragged", scattered(rng, 20000, 40)),
             ("rectangular, wide", scattered(rng, 5000, 25, n_cols=90000)),
             ("rectangular, tall", scattered(rng, 30000, 9, n_cols=700)),
             ("columns in the first third only: empty slabs", scattered(rng, 8000, 30, col_hi=2600)),
             ("one long row among short ones", CRS(2000, np.concatenate([[0], np.cumsum(np.where(np.arange(2000) == 777, 1500, 3))]),
                                                   np.concatenate([np.sort(rng.choice(2000, 1500 if r == 777 else 3, replace=False)) for r in range(2000)]).astype(np.int32),
                                                   rng.uniform(-1, 1, 1500 + 3 * 1999))),
             ("-0.0 and 0.0 products", neg0)]
    ctx.set_option("spmv_valdict", 0)
    ctx.set_option("spmv_win8", 0)
    try:
        for name, A in cases:
            xh = rng.uniform(-2, 2, A.n_cols)
            if A is neg0:
                xh[:] = 1.0
            ctx.set_option("spmv_colslab", 0)
            y0 = _y(ctx, A, xh, want_k=0)
            ctx.set_option("spmv_colslab", K)
            y1 = _y(ctx, A, xh, want_form=7, want_k=K)
            assert np.array_equal(y0.view(np.uint64), y1.view(np.uint64)), name
            yo = oracle.spmv(A, xh)
            scale = np.abs(A.to_scipy()).dot(np.abs(xh)).max() if A.nnz else 1.0
            with np.errstate(over="ignore", invalid="ignore"):
                fin = np.isfinite(yo) & np.isfinite(y1)
                assert np.max(np.abs(y1[fin] - yo[fin]), initial=0.0) <= KTOL * scale, name
    finally:
        for k in ("spmv_valdict", "spmv_win8", "spmv_colslab"):
            ctx.set_option(k, -1)


def test_colslab_refused_for_rows_that_are_not_ascending(ctx, oracle):
    """The reference keeps the input's order inside a row (SURVEY defect 7): a row with descending columns would have its
    sum reordered by the slabs, so the plan is refused and the one-pass kernel answers."""
    rng = np.random.default_rng(7)
    A = scattered(rng, 6000, 20, sort=False)
    xh = rng.uniform(-1, 1, A.n_cols)
    ctx.set_option("spmv_valdict", 0)
    ctx.set_option("spmv_win8", 0)
    ctx.set_option("spmv_colslab", 4)
    try:
        y = _y(ctx, A, xh, want_form=0, want_k=0)
        yo = oracle.spmv(A, xh)
        assert np.max(np.abs(y - yo)) <= KTOL * np.abs(A.to_scipy()).dot(np.abs(xh)).max()
        # ... but rows that are only shuffled INSIDE the column ranges of the slabs keep the plan (the slab index never falls)
        S = scattered(rng, 6000, 20)
        width = -(-S.n_cols // 4)
        col = S.col.copy()
        for r in range(S.n_rows):
            a, z = S.row_ptr[r], S.row_ptr[r + 1]
            for t in range(4):
                idx = a + np.flatnonzero(np.minimum(col[a:z] // width, 3) == t)
                col[idx] = rng.permutation(col[idx])
        S2 = CRS(S.n_rows, S.row_ptr, col, S.val, n_cols=S.n_cols)
        assert (np.diff(col) < 0).sum() > S.n_rows  # (far more descents than row boundaries)
        xs = rng.uniform(-1, 1, S.n_cols)
        ctx.set_option("spmv_colslab", 0)
        y0 = _y(ctx, S2, xs, want_k=0)
        ctx.set_option("spmv_colslab", 4)
        y1 = _y(ctx, S2, xs, want_form=7, want_k=4)
        assert np.array_equal(y0.view(np.uint64), y1.view(np.uint64))
        assert np.max(np.abs(y1 - oracle.spmv(S2, xs))) <= KTOL * np.abs(S2.to_scipy()).dot(np.abs(xs)).max()
        # ... and duplicates of a column (equal, not descending) are fine
        B = CRS(4, [0, 3, 5, 5, 8], np.array([0, 2, 2, 1, 3, 0, 0, 3], dtype=np.int32), np.arange(1.0, 9.0))
        xb = np.array([1.0, -2.0, 0.5, 4.0])
        assert np.allclose(_y(ctx, B, xb, want_form=7, want_k=4), oracle.spmv(B, xb), rtol=1e-14, atol=0)
    finally:
        for k in ("spmv_valdict", "spmv_win8", "spmv_colslab"):
            ctx.set_option(k, -1)


def _spd_scattered(rng, n, per_row):
    """symmetric, strictly diagonally dominant, no locality, ascending rows"""
    import scipy.sparse as sp
    r = np.repeat(np.arange(n), per_row)
    c = rng.integers(0, n, n * per_row)
    M = sp.coo_matrix((rng.uniform(-1, 1, n * per_row), (r, c)), shape=(n, n)).tocsr()
    M = M + M.T
    M.setdiag(0)
    M.eliminate_zeros()
    M = (M + sp.diags(np.asarray(abs(M).sum(axis=1)).ravel() + rng.uniform(1, 2, n))).tocsr()
    M.sort_indices()
    return CRS(n, M.indptr, M.indices.astype(np.int32), M.data)


def test_colslab_in_fused_cg_and_after_scaling(ctx, oracle):
    """The fused (Ap, p) epilogue rides on the LAST pass (methods/cg.hpp:6-54): histories within 1e-10 r0 of the oracle's, like
    the one-pass run; bis_mat_scale_sym drops the slabs and the next SpMV rebuilds them from the new values."""
    rng = np.random.default_rng(11)
    A = _spd_scattered(rng, 12000, 6)
    n = A.n_rows
    ctx.set_option("spmv_valdict", 0)
    ctx.set_option("spmv_win8", 0)
    try:
        ref = oracle.solve(A, "cg", "none")
        for K in (0, 5):
            ctx.set_option("spmv_colslab", K)
            dA = ctx.matrix(A)
            b, x = ctx.alloc(n), ctx.alloc(n)
            ctx.init_vector(b, 1.0); ctx.init_vector(x, 0.1)
            cg = ctx.cg(dA, b, x)
            cg.init(1e-14)
            cg.iterate(60)
            iters, conv, hist = cg.status(hist_cap=128)
            hist = np.array(hist)
            assert dA.colslab_info()[0] == K
            m = min(len(ref["hist"]), len(hist))
            assert m > 3 and np.max(np.abs(ref["hist"][:m] - hist[:m])) <= 1e-10 * ref["hist"][0]
            assert abs(iters - ref["iters"]) <= 1
            cg.free(); b.free(); x.free()
            if K:
                xh = rng.uniform(-1, 1, n)
                dx, dy = ctx.upload(xh), ctx.alloc(n)
                sv = ctx.scale_sym(dA).to_host()
                assert dA.colslab_info()[0] == 0  # dropped with the old values
                ctx.spmv(dA, dx, dy)
                assert dA.colslab_info()[0] == K
                B = CRS(n, A.row_ptr, A.col, A.val * sv[np.repeat(np.arange(n), np.diff(A.row_ptr))] * sv[A.col])
                assert np.max(np.abs(dy.to_host() - oracle.spmv(B, xh))) <= KTOL * np.abs(B.to_scipy()).dot(np.abs(xh)).max()
                dx.free(); dy.free()
            dA.free()
    finally:
        for k in ("spmv_valdict", "spmv_win8", "spmv_colslab"):
            ctx.set_option(k, -1)


def test_colslab_chosen_by_the_trial_on_config5_unstructured(ctx, oracle):
    """`unstr:80,80,80` as generated (config 5 as named: 1.5 M rows numbered at random, 1.04e8 entries): the column stream does
    not pack, x (12.3 MB) does not fit an L2, the rows ascend -> slabs are built and the trial keeps them (measured 1.39 ->
    ~0.55 ms); y equals the one-pass kernel's bit for bit and a sampled slab of the oracle's.  The RCM-ordered matrix has
    locality: it keeps its window form."""
    x = None
    try:
        dA = ctx.gen_unstr(80, 80, 80)
        n = dA.n_rows
        xh = np.random.default_rng(3).uniform(-1, 1, n)
        x, y = ctx.upload(xh), ctx.alloc(n)
        ctx.spmv(dA, x, y)
        K, one_ms, slab_ms = dA.colslab_info()
        assert dA.spmv_stream_info()[3] == 7 and 4 <= K <= 8, (dA.spmv_stream_info(), K)
        assert 0 < slab_ms < 0.85 * one_ms
        y1 = y.to_host()
        ctx.set_option("spmv_colslab", 0)
        dB = ctx.gen_unstr(80, 80, 80)
        ctx.spmv(dB, x, y)
        assert dB.colslab_info()[0] == 0
        assert np.array_equal(y1.view(np.uint64), y.to_host().view(np.uint64))
        ctx.set_option("spmv_colslab", -1)
        rp, col, val = dB.download()
        r0, r1 = 700000, 720000
        S = CRS(r1 - r0, rp[r0:r1 + 1] - rp[r0], col[rp[r0]:rp[r1]], val[rp[r0]:rp[r1]], n_cols=n)
        yo = oracle.spmv(S, xh)
        assert np.max(np.abs(y1[r0:r1] - yo)) <= KTOL * np.max(np.abs(yo))
        perm = ctx.bfs_order(dB, rcm=True)
        dC = ctx.permute(dB, perm)
        ctx.spmv(dC, x, y)
        assert dC.spmv_stream_info()[3] == 6 and dC.colslab_info()[0] == 0
        dA.free(); dB.free(); dC.free(); y.free()
    finally:
        ctx.set_option("spmv_colslab", -1)
        if x is not None:
            x.free()
