"""GPU: the chained sweep (bis_trsv_chain.hip) on LONG triangular rows.

The feeder of a wave pair hands a row to the consumer in 64-entry segments through an 8-slot LDS ring, four segments per
group.  A row of 6 or more segments whose last segment falls into a later group -- or any row of more than 8 segments --
needs the consumer to give slots back segment by segment (round-4 advisor finding: released per ROW, such rows waited for
their own last segment until the spin bound raised BIS_ERR_SYNC).  These tests run rows of 321..1500 strict-triangle
entries at every alignment of the row's first segment within the feeder's groups of four, forward and backward, x aliasing
b included, bit-exact against the oracle's natural-order fma chain (reference kernels.hpp:54-117)."""
import numpy as np
import pytest

from oracle.pyoracle import CRS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from basic_iterative_solvers_amd import Context
    c = Context()
    yield c
    c.close()


def long_row_triangle(n, long_rows, seed, short=5, unlinked_every=0, backward=False, shuffle=False):
    """Strict lower (or, mirrored, upper) triangle in substitution order: every row has its predecessor among its operands
    (one chain, cut by the plan every 128 rows) except every `unlinked_every`-th row, which starts a new chain; rows listed
    in `long_rows` (position -> entries) are long.  Columns ascending unless `shuffle` (the reference keeps the input's
    order inside a row: SURVEY defect 7)."""
    rng = np.random.default_rng(seed)
    rp = [0]
    cols, vals = [], []
    for p in range(n):
        want = min(long_rows.get(p, short), p)
        linked = p > 0 and not (unlinked_every and p % unlinked_every == 0 and p not in long_rows)
        c = set()
        if linked:
            c.add(p - 1)
        lo = 0 if p in long_rows else max(0, p - 400)
        hi = p - 1 if linked else max(p - 40, 0)  # unlinked rows look well behind them (their chain can start early)
        pool = hi - lo
        k = max(0, min(want - len(c), pool))
        if k > 0:
            c.update((lo + rng.choice(pool, size=k, replace=False)).tolist())
        c = np.array(sorted(c), dtype=np.int64)
        if shuffle and len(c) > 1:
            c = rng.permutation(c)
        cols.append(c)
        vals.append(rng.uniform(-1.0, 1.0, len(c)) / max(len(c), 1))
        rp.append(rp[-1] + len(c))
    col = np.concatenate(cols) if cols else np.zeros(0, np.int64)
    val = np.concatenate(vals) if vals else np.zeros(0)
    rp = np.array(rp, dtype=np.int64)
    if backward:  # mirror: position p is row n-1-p, column position q is column n-1-q; rows must be stored ascending by row
        order = []
        new_rp = [0]
        for r in range(n):
            p = n - 1 - r
            order.append(np.arange(rp[p], rp[p + 1]))
            new_rp.append(new_rp[-1] + int(rp[p + 1] - rp[p]))
        idx = np.concatenate(order) if order else np.zeros(0, np.int64)
        col = (n - 1 - col[idx])
        if not shuffle:  # ascending columns inside a row again
            out_c, out_v = [], []
            v2 = val[idx]
            for r in range(n):
                s, e = new_rp[r], new_rp[r + 1]
                o = np.argsort(col[s:e], kind="stable")
                out_c.append(col[s:e][o]); out_v.append(v2[s:e][o])
            col = np.concatenate(out_c); val = np.concatenate(out_v)
        else:
            val = val[idx]
        rp = np.array(new_rp, dtype=np.int64)
    return CRS(n, rp.astype(np.int32), col.astype(np.int32), val.astype(np.float64))


def _run(ctx, oracle, T, backward, chained_expected=True):
    n = T.n_rows
    rng = np.random.default_rng(5)
    D = rng.uniform(1.0, 2.0, n)
    b = rng.uniform(-1, 1, n)
    ref = oracle.sptrsv(T, D, b, backward=backward)
    assert np.all(np.isfinite(ref))
    dT = ctx.matrix(T)
    dD, db, x = ctx.upload(D), ctx.upload(b), ctx.alloc(n)
    solve = ctx.bsptrsv if backward else ctx.sptrsv
    ctx.set_option("trsv_chain", 1)  # the chained sweep wherever its residency bound holds (also below 3 rows per chain)
    try:
        solve(dT, x, dD, db)
        ctx.sync()
        assert np.array_equal(x.to_host(), ref)
        ctx.copy_vector(x, db)  # x aliases b
        solve(dT, x, dD, x)
        assert np.array_equal(x.to_host(), ref)
    finally:
        ctx.set_option("trsv_chain", -1)
    dT.free()


@pytest.mark.parametrize("backward", [False, True])
@pytest.mark.parametrize("align", [0, 1, 2, 3])
def test_chain_long_rows_every_alignment(ctx, oracle, align, backward):
    """Long rows of 6..24 segments whose first segment is the (align)-th of a feeder group: rows before them in the 32-row
    batch are one segment each, so the position of the long row inside its batch sets the alignment."""
    n = 4096
    long_rows = {}
    lens = [321, 330, 384, 385, 449, 512, 513, 600, 777, 1024, 1025, 1500]
    for i, ln in enumerate(lens):
        # chains are cut every 128 rows and a batch is 32 rows: position p = 128 a + 32 b + align is the (align)-th row of a batch
        p = 1600 + 128 * i + 32 * (i % 4) + align
        long_rows[p] = ln
    T = long_row_triangle(n, long_rows, seed=100 + align, backward=backward)
    assert max(np.diff(T.row_ptr)) == 1500
    _run(ctx, oracle, T, backward)


@pytest.mark.parametrize("backward", [False, True])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_chain_long_rows_randomised(ctx, oracle, seed, backward):
    """Random positions and lengths (65..1500), several long rows in one batch and back to back, chains that start at rows
    without a predecessor link (several chains in flight), unsorted columns on one seed."""
    rng = np.random.default_rng(seed)
    n = 6000
    long_rows = {}
    for p in rng.choice(np.arange(1600, n), size=40, replace=False):
        long_rows[int(p)] = int(rng.integers(65, 1501))
    p0 = 3000
    for k in range(6):  # six long rows back to back
        long_rows[p0 + k] = int(rng.integers(321, 900))
    T = long_row_triangle(n, long_rows, seed=seed, unlinked_every=37, backward=backward, shuffle=(seed == 3))
    _run(ctx, oracle, T, backward)


def test_chain_rows_longer_than_the_ring(ctx, oracle):
    """[10, 512+, ...]: a row of more than 8 segments right behind a short one (the advisor's model case)."""
    n = 3000
    T = long_row_triangle(n, {2001: 600, 2002: 10, 2003: 513, 2500: 1400}, seed=9)
    _run(ctx, oracle, T, False)
