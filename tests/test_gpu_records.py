"""GPU: what a record of a run is made of (round 5): the options in effect (bis_options_describe), the HIP-event bracket around
every sweep call (bis_profile_read_sweeps) and the name of the kernel a sweep ran (bis_mat_sweep_kernel) -- the run-time
counterparts of the reference's compile-time configuration (CMakeLists.txt:19-29) and LIKWID regions (kernels.hpp:56-58, :90-92)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from basic_iterative_solvers_amd import Context
    c = Context()
    yield c
    c.close()


def test_options_in_effect_are_reported(ctx):
    base = ctx.options()
    assert "env" in base and "spmv_valdict" not in base
    ctx.set_option("spmv_valdict", 0)
    ctx.set_option("trsv_chain_pairs", 40)
    try:
        o = ctx.options()
        assert o["spmv_valdict"] == 0 and o["trsv_chain_pairs"] == 40
    finally:
        ctx.set_option("spmv_valdict", -1)
        ctx.set_option("trsv_chain_pairs", -1)
    assert ctx.options() == base
    with pytest.raises(Exception):
        ctx.set_option("no_such_option", 1)


def test_sweep_profile_and_kernel_names(ctx, oracle):
    """Every bis_sptrsv / bis_bsptrsv call is one bracket (all its launches); the triangle remembers which kernel served it: the
    tiled sweep on a grid-hinted matrix, a level-scheduled kernel where the tiled sweep is switched off, the chained sweep on
    a banded matrix without a grid."""
    n1 = 24
    dA = ctx.gen_hpcg(n1)
    n = dA.n_rows
    Ls, Us, D, Dinv = ctx.split_strict(dA)
    b, x = ctx.upload(np.random.default_rng(2).uniform(-1, 1, n)), ctx.alloc(n)
    assert Ls.sweep_kernel(False) == "" and Us.sweep_kernel(True) == ""
    ctx.profile(True)
    for _ in range(3):
        ctx.sptrsv(Ls, x, D, b)
    ctx.bsptrsv(Us, x, D, b)
    ctx.sync()
    ctx.profile(False)
    k, ms = ctx.profile_read_sweeps()
    assert k == 4 and ms > 0
    assert ctx.profile_read_sweeps() == (0, 0.0)  # read once
    assert Ls.sweep_kernel(False) == "trsv_tiled_kernel" and Us.sweep_kernel(True) == "trsv_tiled_kernel"
    ctx.sptrsv(Ls, x, D, b)  # not profiled: no bracket
    ctx.sync()
    assert ctx.profile_read_sweeps()[0] == 0
    ctx.set_option("trsv_tiled", 0)
    ctx.set_option("trsv_chain", 0)
    try:
        L2, U2, D2, Di2 = ctx.split_strict(dA)
        ctx.sptrsv(L2, x, D2, b)
        ctx.sync()
        assert L2.sweep_kernel(False) in ("sptrsv_wave_kernel", "sptrsv_syncfree_kernel")
        for m in (L2, U2):
            m.free()
    finally:
        ctx.set_option("trsv_tiled", -1)
        ctx.set_option("trsv_chain", -1)
    # a banded matrix without a grid: the chained sweep
    from oracle.pyoracle import CRS
    nb, w = 20000, 6
    lens = np.minimum(np.arange(nb), w)
    rp = np.concatenate([[0], np.cumsum(lens)])
    col = np.concatenate([np.arange(r - lens[r], r) for r in range(nb)]).astype(np.int32)
    val = np.random.default_rng(4).uniform(-1, 1, rp[-1]) / w
    T = ctx.matrix(CRS(nb, rp, col, val))
    Db, bb, xb = ctx.upload(np.full(nb, 2.0)), ctx.upload(np.random.default_rng(5).uniform(-1, 1, nb)), ctx.alloc(nb)
    ctx.sptrsv(T, xb, Db, bb)
    ctx.sync()
    assert T.sweep_kernel(False) == "trsv_chain_kernel"
    want = oracle.sptrsv(CRS(nb, rp, col, val), np.full(nb, 2.0), bb.to_host())
    assert np.array_equal(xb.to_host(), want)
    for m in (Ls, Us, dA, T):
        m.free()
