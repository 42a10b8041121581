"""The 64-bit row-pointer code path -- what HPCG-512 (3.6e9 non-zeros, the
north-star target size; not representable in the reference's int CRS,
sparse_matrix.hpp:60-66) runs on -- under the same parity tests as the 32-bit
path: `force_rp64` makes every matrix created through the C ABI carry int64
row pointers, so the RP = int64_t instantiations of the SpMV, both triangular
sweeps, the strict split, ILU(0), the generators, the level analysis, the
reordering and the fused CG run on the golden inputs."""
import numpy as np
import pytest

import test_gpu_kernels as T
from helpers import GOLDEN_MATS, crs_of, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from basic_iterative_solvers_amd import Context
    c = Context()
    c.set_option("force_rp64", 1)
    probe = c.gen_hpcg(4)
    assert probe.rp_width == 8, "force_rp64 not honoured by the generators"
    probe.free()
    g = load_golden("hpcg8")
    m = c.matrix(crs_of(g, "A"))
    assert m.rp_width == 8, "force_rp64 not honoured by bis_mat_create"
    Ls, Us, D, Dinv = c.split_strict(m)
    assert Ls.rp_width == 8 and Us.rp_width == 8, "strict split lost the 64-bit row pointers"
    yield c
    c.set_option("force_rp64", -1)
    c.close()


@pytest.mark.parametrize("name", GOLDEN_MATS)
def test_rp64_kernels_vs_reference_golden(ctx, name):
    T.test_kernels_vs_reference_golden(ctx, name)


@pytest.mark.parametrize("name", GOLDEN_MATS)
def test_rp64_split_strict_bit_exact(ctx, name):
    T.test_split_strict_bit_exact(ctx, name)


@pytest.mark.parametrize("name", GOLDEN_MATS)
def test_rp64_device_ilu0_vs_reference_factors(ctx, name):
    T.test_device_ilu0_vs_reference_factors(ctx, name)


@pytest.mark.parametrize("kind,size", [("hpcg", 32), ("anderson", 40)])
def test_rp64_medium_size_vs_oracle(ctx, oracle, kind, size):
    T.test_medium_size_vs_oracle(ctx, oracle, kind, size)


def test_rp64_generators_bit_exact(ctx, oracle):
    T.test_hpcg_generator_bit_exact(ctx, oracle, (17, 9, 11))
    T.test_anderson_generator_bit_exact(ctx, oracle, 13, 0.0)
    T.test_fem_generator_bit_exact(ctx, oracle, (5, 4, 7), 60, (37, 301))


def test_rp64_fem_kernels_and_ilu0(ctx, oracle):
    T.test_fem_medium_kernels_and_ilu0_vs_oracle(ctx, oracle)


def test_rp64_spmv_ragged_and_long_rows(ctx, oracle):
    T.test_spmv_ragged_unsorted_and_empty_rows(ctx, oracle)
    T.test_spmv_very_long_rows_fallback(ctx, oracle)
    T.test_spmv_degenerate_shapes(ctx, oracle)


def test_rp64_sptrsv_few_level_path(ctx, oracle):
    T.test_sptrsv_few_level_path_bit_exact(ctx, oracle)


@pytest.mark.parametrize("kind", ["hpcg", "fem", "klein"])
def test_rp64_multicolour(ctx, oracle, kind):
    T.test_device_multicolour_reordering(ctx, oracle, kind)


@pytest.mark.parametrize("name", ["hpcg8", "matrix_band_klein"])
def test_rp64_scale_sym(ctx, oracle, name):
    T.test_device_scale_sym_bit_exact(ctx, oracle, name)


@pytest.mark.parametrize("key", [k for k in T._CG_KEYS if k.split("|")[0] in ("hpcg8", "FDM-2d-16", "anderson8_shift9")])
def test_rp64_fused_cg_history_vs_reference(ctx, key):
    T.test_fused_cg_history_vs_reference(ctx, key)


def test_rp64_download_round_trip(ctx):
    g = load_golden("matrix_band_klein")
    A = crs_of(g, "A")
    m = ctx.matrix(A)
    rp, col, val = m.download()
    assert np.array_equal(rp, A.row_ptr) and np.array_equal(col, A.col) and np.array_equal(val, A.val)
