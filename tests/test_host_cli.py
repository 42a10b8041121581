"""GPU: the C++ host layer (basic_iterative_solvers_amd/host, the reference's
CLI and solver classes on top of the C ABI) against the reference's residual
tables for every solver x preconditioner pair in tests/golden/histories.json.
The stdout residual table is the parity artefact (postprocessing.hpp:8-30)."""
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, check_history, load_histories, load_histories_mid, load_histories_r4, parse_hist_key

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "basic_iterative_solvers_amd", "host", "basic_iterative_solvers")

MATRIX_ARG = {
    "FDM-2d-16": os.path.join(GOLDEN, "FDM-2d-16.mtx"),
    "matrix_band_klein": os.path.join(GOLDEN, "matrix_band_klein.mtx"),
    "hpcg8": "hpcg:8",
    "hpcg_4x6x5": "hpcg:4,6,5",
    "anderson8_shift9": "anderson:8,shift=9",
    "hpcg176": "hpcg:176",
}

_H = load_histories()
RES = re.compile(r"\|\|A\*x_(\d+) - b\|\|_2 = (\S+)")


def run_cli(name, solver, pc, kw, extra=()):
    cmd = [BIN, MATRIX_ARG[name], "-" + solver]
    if pc != "none":
        cmd += ["-p", pc]
    if kw.get("num_scale"):
        cmd += ["-scale", "1"]
    if "restart_len" in kw:
        cmd += ["-rl", str(kw["restart_len"])]
    cmd += list(extra)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    hist = [float(v) for _, v in RES.findall(out.stdout)]
    m = re.search(r"(converged in: |did not converge after )(\d+) iterations", out.stdout)
    assert m, out.stdout[-1500:]
    return dict(hist=np.array(hist), iters=int(m.group(2)), converged=m.group(1).startswith("converged"),
                stdout=out.stdout)


@pytest.mark.parametrize("key", sorted(_H))
def test_cli_residual_table_vs_reference(key):
    assert os.path.exists(BIN), "host binary not built (make -C basic_iterative_solvers_amd/host)"
    e = _H[key]
    name, solver, pc, kw = parse_hist_key(key)
    r = run_cli(name, solver, pc, kw)
    assert abs(r["hist"][0] - e["hist"][0]) <= 1e-13 * e["hist"][0]
    check_history(r, e, solver, stable_window=True)
    if solver in ("cg", "j", "gs", "sgs") and e["iters"] is not None and len(r["hist"]) == len(e["hist"]):
        assert r["iters"] == e["iters"]


_HM = load_histories_mid()


@pytest.mark.parametrize("key", sorted(_HM))
def test_cli_mid_size_history_vs_reference(key):
    """Full residual histories of the REAL reference (oracle/_ref at one thread, tests/golden/histories_mid.json) on
    inputs where row blocks, x windows, sweep tiles and levels are many (HPCG-48: 110,592 rows; HPCG-32; Anderson-32
    shift 9; the FEM stand-in 12x11x10), against the HIP path on the device-generated twin of the same input, to
    convergence (solver_harness.hpp:7-61): max_k |r_k - r_k^ref| <= 1e-10 r0 (BiCGSTAB: first iterations at 1e-10,
    whole history 1e-4, as for the small inputs), same iteration count."""
    e = _HM[key]
    name, solver, pc, kw = parse_hist_key(key)
    MATRIX_ARG[name] = e["cli"]
    r = run_cli(name, solver, pc, kw)
    # (r0 is a sum of 10^5 squares: the reference's sequential sum and the device's tree agree to 1.1e-13 on HPCG-48)
    assert abs(r["hist"][0] - e["hist"][0]) <= 1e-12 * e["hist"][0]
    check_history(r, e, solver, stable_window=True, long_history=True)
    if solver in ("cg", "j", "gs", "sgs") and len(r["hist"]) == len(e["hist"]):
        assert r["iters"] == e["iters"]


_H4 = load_histories_r4()


@pytest.mark.parametrize("key", sorted(_H4))
def test_cli_round4_histories_vs_reference(key):
    """(i) BASELINE configs 2-3 as named run CG on the RAW Anderson operator (indefinite): the first 100 iterations of the
    reference (cg.hpp:6-54; oracle/_ref at one thread with max_iters = 100) against the first 100 of the CLI's run --
    SURVEY 8d's parity gate item (ii); the Jacobi-preconditioned run over the window in which the reference itself is
    independent of rounding (`stable_len`).  (ii) Config 5's unstructured input (`unstr:`: the FEM-like matrix under a
    random row permutation -- no grid hint, so the sweeps and ILU(0) take the general kernels) with the solver /
    preconditioner pairs of configs 5 and 4, to convergence."""
    e = _H4[key]
    name, solver, pc, kw = parse_hist_key(key)
    MATRIX_ARG[name] = e["cli"]
    kw = dict(kw)
    raw = kw.pop("max_iters", None)
    r = run_cli(name, solver, pc, kw)
    assert abs(r["hist"][0] - e["hist"][0]) <= 1e-12 * e["hist"][0]
    if raw is not None:
        n = min(len(e["hist"]), e.get("stable_len", len(e["hist"])))
        assert len(r["hist"]) >= n
        g, h = np.array(e["hist"][:n]), r["hist"][:n]
        # (a window cut by `stable_len` is by definition the part on which differently rounded runs agree to 1e-9 r0)
        # (plain CG: 1e-10 r0; Jacobi on the indefinite diagonal sends the residual to 10^3 r0 inside the window: 1e-10 of its largest value)
        assert np.max(np.abs(h - g)) <= (1e-10 * g[0] if n == len(e["hist"]) else 1e-10 * np.max(g))
        return
    check_history(r, e, solver, stable_window=True)
    if solver in ("cg", "j", "gs", "sgs") and len(r["hist"]) == len(e["hist"]):
        assert r["iters"] == e["iters"]


@pytest.mark.parametrize("mode", ["rcm", "bfs"])
@pytest.mark.parametrize("solver,pc", [("bi", "ilu0"), ("gm", "gs")])
def test_cli_unstructured_input_reordered_vs_oracle_on_permuted_matrix(tmp_path, oracle, mode, solver, pc):
    """The realistic pipeline for an unstructured mesh: `unstr:` (no locality at all) -> RCM / BFS on the device -> ILU(0)
    / Gauss-Seidel sweeps in that order.  Parity target: the reference algorithm (oracle, real ILU(0)) on the same
    permuted matrix P A P^T (the permutation is read back with -dump-perm)."""
    from helpers import permute_crs
    MATRIX_ARG["unstr_9x8x7"] = "unstr:9,8,7,seed=5"
    permfile = str(tmp_path / "perm.txt")
    kw = {"restart_len": 50} if solver == "gm" else {}
    r = run_cli("unstr_9x8x7", solver, pc, kw, extra=["-perm", mode, "-dump-perm", permfile])
    perm = np.loadtxt(permfile, dtype=np.int64)
    A = oracle.gen_unstr(9, 8, 7, seed=5)
    assert sorted(perm) == list(range(A.n_rows))
    B = permute_crs(A, perm)
    if mode == "rcm":  # the point of RCM on a matrix without locality: the bandwidth collapses
        bw = lambda M: int(np.max(np.abs(M.col.astype(np.int64) - np.repeat(np.arange(M.n_rows), np.diff(M.row_ptr)))))
        assert bw(B) < bw(A) // 3
    o = oracle.solve(B, solver, pc, ilu_real=True, **kw)
    e = dict(hist=[float(v) for v in o["hist"]], iters=o["iters"], converged=o["converged"])
    check_history(r, e, solver)
    if solver != "bi":
        assert abs(r["iters"] - o["iters"]) <= 1


@pytest.mark.parametrize("solver", ["j", "gs", "sgs"])
@pytest.mark.parametrize("name", ["FDM-2d-16", "matrix_band_klein", "hpcg8", "anderson8_shift9"])
def test_cli_stationary_device_schedule_equals_unfused(name, solver):
    """-j / -gs / -sgs run the device schedule (bis_stat_*: norm and stopping test on the device, Jacobi with one
    SpMV per iteration) by default; -unfused is the reference's kernel-by-kernel order with a blocking norm per
    iteration (jacobi.hpp:43-52,:102-107; gauss_seidel.hpp:26-52,:99-104).  Same arithmetic per element and per
    partial sum: the printed residual tables are identical digit for digit, so is the iteration count (also
    where the solver does not converge: FDM-2d-16 -j runs to MAX_ITERS in the reference)."""
    a = run_cli(name, solver, "none", {})
    b = run_cli(name, solver, "none", {}, extra=["-unfused"])
    assert a["iters"] == b["iters"] and a["converged"] == b["converged"]
    assert len(a["hist"]) == len(b["hist"]) and np.array_equal(a["hist"], b["hist"])


@pytest.mark.parametrize("solver", ["j", "gs"])
def test_cli_stationary_device_schedule_equals_unfused_above_the_grid_caps(solver):
    """The same identity at 5.45 M rows (HPCG 176^3): above 8192 x 512 elements the reductions' grids sit at their cap
    (kMaxDotBlocks), and the Jacobi step's fused residual norm must share the index map of the stand-alone dot to stay
    digit-identical with the kernel-by-kernel schedule."""
    a = run_cli("hpcg176", solver, "none", {})
    b = run_cli("hpcg176", solver, "none", {}, extra=["-unfused"])
    assert a["iters"] == b["iters"] and a["converged"] == b["converged"]
    assert len(a["hist"]) == len(b["hist"]) and np.array_equal(a["hist"], b["hist"])


@pytest.mark.parametrize("pc", ["none", "j"])
def test_cli_fused_and_unfused_cg_agree(pc):
    """The fused device schedule and the reference's kernel-by-kernel order
    print the same table to 1e-12 r0."""
    a = run_cli("hpcg8", "cg", pc, {})
    b = run_cli("hpcg8", "cg", pc, {}, extra=["-unfused"])
    assert a["iters"] == b["iters"]
    assert np.max(np.abs(a["hist"] - b["hist"])) <= 1e-12 * a["hist"][0]


def test_cli_output_layout():
    r = run_cli("FDM-2d-16", "cg", "none", {})
    s = r["stdout"]
    assert "               Residual Norms                           Time for iteration" in s
    assert "Solver: conjugate-gradient converged in: 34 iterations." in s
    assert 'With the stopping criteria "tol * ||Ax_0 - b||_2" is: ' in s
    assert "The residual of the final iteration is: ||A*x_star - b||_2 = " in s
    assert "| | | SpMV time: " in s and "Total elapsed time: " in s
    assert "res3 => iter_count: " in s and "res6 => iter_count: " in s


def test_cli_errors():
    out = subprocess.run([BIN, os.path.join(GOLDEN, "nope.mtx"), "-cg"], capture_output=True, text=True)
    assert out.returncode != 0 and "Unable to open file" in out.stderr
    out = subprocess.run([BIN, "hpcg:4"], capture_output=True, text=True)
    assert out.returncode != 0 and "Not enough arguments" in out.stdout


@pytest.mark.parametrize("solver,pc", [("bi", "ilu0"), ("cg", "j"), ("gm", "gs"), ("cg", "sgs"), ("gs", "none")])
def test_cli_fem_generator_histories_vs_oracle(oracle, solver, pc):
    """Config 5's input family (FEM-like unstructured generator, `fem:NX,NY,NZ`):
    the CLI's residual table against the reference algorithm (oracle, real ILU(0))
    on the oracle's own copy of the matrix."""
    MATRIX_ARG["fem_6x5x4"] = "fem:6,5,4,seed=3"
    kw = {"restart_len": 30} if solver == "gm" else {}
    r = run_cli("fem_6x5x4", solver, pc, kw)
    A = oracle.gen_fem(6, 5, 4, seed=3)
    o = oracle.solve(A, solver, pc, ilu_real=True, **kw)
    e = dict(hist=[float(v) for v in o["hist"]], iters=o["iters"], converged=o["converged"])
    assert abs(r["hist"][0] - e["hist"][0]) <= 1e-13 * e["hist"][0]
    check_history(r, e, solver)
    if solver != "bi":
        assert abs(r["iters"] - o["iters"]) <= 1


@pytest.mark.parametrize("name,solver,pc", [("hpcg8", "gm", "gs"), ("hpcg8", "cg", "sgs"),
                                            ("anderson8_shift9", "bi", "ilu0"),
                                            ("anderson8_shift9", "gs", "none"),
                                            ("FDM-2d-16", "cg", "sgs")])
def test_cli_multicolour_reordering_vs_oracle_on_permuted_matrix(tmp_path, oracle, name, solver, pc):
    """-perm mc changes the Gauss-Seidel / ILU iteration; its parity target is
    the reference algorithm (oracle) run on the same permuted matrix P A P^T
    (the protocol of the reference's SMAX path, smax_helpers.hpp:44-80)."""
    from helpers import crs_of, load_golden
    from oracle.pyoracle import CRS
    permfile = str(tmp_path / "perm.txt")
    kw = {"restart_len": 50} if solver == "gm" else {}
    r = run_cli(name, solver, pc, kw, extra=["-perm", "mc", "-dump-perm", permfile])
    perm = np.loadtxt(permfile, dtype=np.int64)
    A = crs_of(load_golden(name), "A")
    n = A.n_rows
    assert sorted(perm) == list(range(n))
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    lens = np.diff(A.row_ptr)[perm]
    rp = np.concatenate([[0], np.cumsum(lens)])
    col = np.concatenate([inv[A.col[A.row_ptr[o]:A.row_ptr[o + 1]]] for o in perm]).astype(np.int32)
    val = np.concatenate([A.val[A.row_ptr[o]:A.row_ptr[o + 1]] for o in perm])
    B = CRS(n, rp, col, val)
    # colouring property: no row couples to a row of its own colour block -> few levels
    m = re.search(r"multi-colour reordering: (\d+) colours", r["stdout"])
    assert m and int(m.group(1)) <= 16
    o = oracle.solve(B, solver, pc, ilu_real=True, **kw)
    e = dict(hist=[float(v) for v in o["hist"]], iters=o["iters"], converged=o["converged"])
    check_history(r, e, solver)
    if solver != "bi":
        assert abs(r["iters"] - o["iters"]) <= 1


@pytest.mark.parametrize("name,solver,pc,scale", [("FDM-2d-16", "cg", "sgs", True), ("matrix_band_klein", "gs", "none", True),
                                                  ("hpcg_4x6x5", "gm", "gs", False), ("anderson8_shift9", "bi", "ilu0", True)])
def test_cli_device_and_host_reordering_agree(tmp_path, name, solver, pc, scale):
    """-perm mc on the device (bis_mat_multicolour, bis_vec_gather for the rescaled b) and the
    host fallback (-perm-host) build the same permutation and the same P A P^T: identical
    permutation files and residual tables that agree to rounding of the device reductions."""
    kw = {"num_scale": True} if scale else {}
    if solver == "gm":
        kw["restart_len"] = 30
    f1, f2 = str(tmp_path / "p_dev.txt"), str(tmp_path / "p_host.txt")
    a = run_cli(name, solver, pc, kw, extra=["-perm", "mc", "-dump-perm", f1])
    b = run_cli(name, solver, pc, kw, extra=["-perm", "mc", "-perm-host", "-dump-perm", f2])
    assert np.array_equal(np.loadtxt(f1, dtype=np.int64), np.loadtxt(f2, dtype=np.int64))
    assert a["iters"] == b["iters"] and len(a["hist"]) == len(b["hist"])
    assert np.max(np.abs(a["hist"] - b["hist"])) <= 1e-12 * a["hist"][0]


@pytest.mark.parametrize("mode,name,solver,pc", [("rcm", "FDM-2d-16", "cg", "sgs"), ("bfs", "matrix_band_klein", "gs", "none"),
                                                 ("rcm", "hpcg_4x6x5", "gm", "gs"), ("bfs", "anderson8_shift9", "bi", "ilu0")])
def test_cli_rcm_bfs_reordering_vs_oracle_on_permuted_matrix(tmp_path, oracle, mode, name, solver, pc):
    """-perm rcm / -perm bfs (SMAX PERM_MODE roles): any valid permutation is accepted; the parity
    target is the reference algorithm (oracle) on the same permuted matrix P A P^T."""
    from helpers import crs_of, load_golden
    from oracle.pyoracle import CRS
    permfile = str(tmp_path / "perm.txt")
    kw = {"restart_len": 50} if solver == "gm" else {}
    r = run_cli(name, solver, pc, kw, extra=["-perm", mode, "-dump-perm", permfile])
    perm = np.loadtxt(permfile, dtype=np.int64)
    A = crs_of(load_golden(name), "A")
    n = A.n_rows
    assert sorted(perm) == list(range(n))
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    lens = np.diff(A.row_ptr)[perm]
    rp = np.concatenate([[0], np.cumsum(lens)])
    col = np.concatenate([inv[A.col[A.row_ptr[o]:A.row_ptr[o + 1]]] for o in perm]).astype(np.int32)
    val = np.concatenate([A.val[A.row_ptr[o]:A.row_ptr[o + 1]] for o in perm])
    B = CRS(n, rp, col, val)
    if mode == "rcm":  # the point of RCM: the bandwidth does not grow on these banded inputs
        bw = lambda M: max(abs(int(c) - r_) for r_ in range(M.n_rows) for c in M.col[M.row_ptr[r_]:M.row_ptr[r_ + 1]])
        assert bw(B) <= bw(A) * 2
    o = oracle.solve(B, solver, pc, ilu_real=True, **kw)
    e = dict(hist=[float(v) for v in o["hist"]], iters=o["iters"], converged=o["converged"])
    check_history(r, e, solver)
    if solver != "bi":
        assert abs(r["iters"] - o["iters"]) <= 1


def test_cli_binary_crs_cache(tmp_path):
    """-cache FILE: the first run parses the .mtx and writes the binary CRS, the second reads it;
    identical residual tables."""
    cache = str(tmp_path / "fdm.crs")
    a = run_cli("FDM-2d-16", "cg", "j", {}, extra=["-cache", cache])
    assert os.path.getsize(cache) > 1000
    b = run_cli("FDM-2d-16", "cg", "j", {}, extra=["-cache", cache])
    assert np.array_equal(a["hist"], b["hist"]) and a["iters"] == b["iters"]
    # a cache file that is not one is ignored (falls back to the parser)
    with open(cache, "wb") as f:
        f.write(b"not a cache")
    c = run_cli("FDM-2d-16", "cg", "j", {}, extra=["-cache", cache])
    assert np.array_equal(a["hist"], c["hist"])


def test_cli_is_linked_with_roctx():
    """The reference's LIKWID markers around spmv / sptrsv / backwards-sptrsv are roctx ranges here."""
    out = subprocess.run(["ldd", BIN], capture_output=True, text=True)
    assert "libroctx64" in out.stdout, out.stdout


@pytest.mark.parametrize("mode,extra", [("mc", []), ("mc", ["-perm-host"]), ("rcm", []), ("bfs", [])])
@pytest.mark.parametrize("name,solver,pc", [("hpcg_4x6x5", "cg", "none"), ("anderson8_shift9", "cg", "sgs"),
                                            ("FDM-2d-16", "cg", "j")])
def test_cli_perm_returns_x_star_in_natural_order(tmp_path, mode, extra, name, solver, pc):
    """-perm solves P A P^T (Px) = P b; x* is handed back in the caller's row order (the reference's
    SMAX path leaves it permuted, smax_helpers.hpp:44-80): it equals the x* of the un-permuted solve
    to the accuracy both solves reach (stopping tolerance 1e-14 r0), not a permutation of it."""
    f0, f1, fp = str(tmp_path / "x0.txt"), str(tmp_path / "x1.txt"), str(tmp_path / "perm.txt")
    a = run_cli(name, solver, pc, {}, extra=["-dump-x", f0])
    b = run_cli(name, solver, pc, {}, extra=["-perm", mode, "-dump-perm", fp, "-dump-x", f1] + extra)
    assert a["converged"] and b["converged"]
    x0, x1 = np.loadtxt(f0), np.loadtxt(f1)
    perm = np.loadtxt(fp, dtype=np.int64)
    assert not np.array_equal(perm, np.arange(len(perm)))  # the permutation is not trivial ...
    scale = np.max(np.abs(x0))
    assert np.max(np.abs(x1 - x0)) <= 1e-9 * scale         # ... and x* is back in natural order
    assert np.max(np.abs(x1[perm] - x0)) > 1e-6 * scale    # (the still-permuted vector would not pass)


@pytest.mark.parametrize("name,solver,pc,kw", [("hpcg8", "gm", "gs", {"restart_len": 10}), ("FDM-2d-16", "gm", "none", {"restart_len": 30}),
                                               ("anderson8_shift9", "bi", "ilu0", {}), ("hpcg_4x6x5", "bi", "sgs", {}),
                                               ("matrix_band_klein", "bi", "j", {})])
def test_cli_device_scalar_schedules_match_host_scalar_ones(name, solver, pc, kw):
    """GMRES keeps its Gram-Schmidt coefficients, BiCGSTAB its rho / alpha / omega / beta on the device (one blocking
    read per iteration instead of j+2 / 6); `-hostscalars` returns every dot product to the host like the reference.
    Same kernels, same IEEE operations in the same order: the printed residual tables are identical digit for digit."""
    a = run_cli(name, solver, pc, kw)
    b = run_cli(name, solver, pc, kw, extra=["-hostscalars"])
    assert a["iters"] == b["iters"] and a["converged"] == b["converged"]
    assert np.array_equal(a["hist"], b["hist"])


@pytest.mark.parametrize("mode,name,solver,pc,scale", [("rcm", "FDM-2d-16", "cg", "sgs", True), ("bfs", "hpcg_4x6x5", "gm", "gs", False),
                                                       ("rcm", "anderson8_shift9", "bi", "ilu0", True), ("bfs", "matrix_band_klein", "gs", "none", False)])
def test_cli_device_and_host_rcm_bfs_agree(tmp_path, mode, name, solver, pc, scale):
    """-perm rcm|bfs on the device (bis_mat_bfs_order + bis_mat_permute) and the sequential host version (-perm-host)
    produce the same permutation file and residual tables that agree to rounding of the device reductions."""
    kw = {"num_scale": True} if scale else {}
    if solver == "gm":
        kw["restart_len"] = 30
    f1, f2 = str(tmp_path / "p_dev.txt"), str(tmp_path / "p_host.txt")
    a = run_cli(name, solver, pc, kw, extra=["-perm", mode, "-dump-perm", f1])
    b = run_cli(name, solver, pc, kw, extra=["-perm", mode, "-perm-host", "-dump-perm", f2])
    assert "(host)" not in a["stdout"] and "(host)" in b["stdout"]
    assert np.array_equal(np.loadtxt(f1, dtype=np.int64), np.loadtxt(f2, dtype=np.int64))
    assert a["iters"] == b["iters"] and len(a["hist"]) == len(b["hist"])
    assert np.max(np.abs(a["hist"] - b["hist"])) <= 1e-12 * a["hist"][0]


@pytest.mark.parametrize("solver,pc", [("cg", "sgs"), ("gm", "gs"), ("bi", "ilu0"), ("gs", "none")])
def test_cli_grid_hint_and_trsv_modes(tmp_path, oracle, monkeypatch, solver, pc):
    """-grid NX,NY,NZ[,DOF] tells the triangular sweeps that a matrix read from a file is a stencil on a grid (here
    the 27-point pattern on 12 x 13 x 14 written as general MatrixMarket): they then run tiled (plan built on the
    device), and since every row keeps the reference's CRS-order fma chain (kernels.hpp:54-107) the printed residual
    table is the same, digit for digit, as with the level-scheduled kernels (-trsv level) and as without the hint.
    A generated matrix carries the hint by itself."""
    monkeypatch.setenv("BIS_TRSV_TILE_STATS", "1")
    A = oracle.gen_hpcg(12, 13, 14)
    mtx = str(tmp_path / "grid.mtx")
    with open(mtx, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (A.n_rows, A.n_rows, len(A.col)))
        rows = np.repeat(np.arange(A.n_rows), np.diff(A.row_ptr))
        for r, c, v in zip(rows, A.col, A.val):
            f.write("%d %d %.17g\n" % (r + 1, c + 1, v))

    def run(matrix, extra):
        cmd = [BIN, matrix, "-" + solver] + (["-p", pc] if pc != "none" else []) + extra
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        return RES.findall(out.stdout), out.stderr.count("tiled sptrsv plan"), out.stderr.count("device plan")

    plain, n_plain, _ = run(mtx, [])
    hinted, n_hinted, n_dev = run(mtx, ["-grid", "12,13,14"])
    level, n_level, _ = run(mtx, ["-grid", "12,13,14", "-trsv", "level"])
    assert len(plain) > 3 and plain == hinted == level
    assert n_plain == 0 and n_level == 0 and n_hinted >= 1 and n_dev == n_hinted
    gen, n_gen, n_gen_dev = run("hpcg:12,13,14", [])
    gen_level, n_gl, _ = run("hpcg:12,13,14", ["-trsv", "level"])
    assert gen == gen_level == plain and n_gen >= 1 and n_gen_dev == n_gen and n_gl == 0
    bad = subprocess.run([BIN, mtx, "-cg", "-grid", "12,13,13"], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "-grid" in bad.stderr
    # from 4096 rows on bis_mat_create recognises the grid of such a file by itself (column offsets of a few rows)
    B = oracle.gen_hpcg(16, 17, 18)
    mtx2 = str(tmp_path / "grid2.mtx")
    with open(mtx2, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (B.n_rows, B.n_rows, len(B.col)))
        rows = np.repeat(np.arange(B.n_rows), np.diff(B.row_ptr))
        np.savetxt(f, np.column_stack([rows + 1, B.col + 1, B.val]), fmt="%d %d %.17g")
    auto, n_auto, n_auto_dev = run(mtx2, [])
    lvl, n_lvl, _ = run(mtx2, ["-trsv", "level"])
    assert len(auto) > 3 and auto == lvl and n_auto >= 1 and n_auto_dev == n_auto and n_lvl == 0


def test_cli_mid_size_mtx_file_equals_generator_and_reference(tmp_path, oracle):
    """The `.mtx` input path at a size where it matters (sparse_matrix.hpp:225-357: 830 584 entries, 17 MB of text, the
    threaded reader): HPCG-32 written as a general MatrixMarket file with its ROWS in shuffled order (entries of a row
    stay in ascending column order -- the reference's reader keeps the file's order inside a row) gives the same CRS,
    hence the same residual table digit for digit as the generator input `hpcg:32`, and the reference's own history
    (tests/golden/histories_mid.json, `hpcg32|sgs`) to 1e-10 r0; the binary CRS cache written on the way reproduces it."""
    A = oracle.gen_hpcg(32)
    rng = np.random.default_rng(6)
    order = rng.permutation(A.n_rows)
    lens = np.diff(A.row_ptr)
    idx = np.concatenate([np.arange(A.row_ptr[r], A.row_ptr[r + 1]) for r in order])
    rows = np.repeat(order, lens[order])
    mtx = str(tmp_path / "HPCG-32.mtx")
    with open(mtx, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real general\n%% rows in shuffled order\n%d %d %d\n" % (A.n_rows, A.n_rows, A.nnz))
        np.savetxt(f, np.column_stack([rows + 1, A.col[idx] + 1, A.val[idx]]), fmt="%d %d %.17g")

    def run(matrix, extra=()):
        out = subprocess.run([BIN, matrix, "-sgs"] + list(extra), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        return [float(v) for _, v in RES.findall(out.stdout)]

    gen = run("hpcg:32")
    cache = str(tmp_path / "hpcg32.crs")
    from_file = run(mtx, ["-cache", cache])
    assert len(gen) > 100 and from_file == gen
    assert os.path.exists(cache) and run(mtx, ["-cache", cache]) == gen
    e = _HM["hpcg32|sgs|none|"]
    check_history(dict(hist=np.array(from_file), iters=len(from_file) - 1, converged=True), e, "sgs", long_history=True)


@pytest.mark.parametrize("symmetric", [0, 1])
def test_cli_mtx_ingestion_general_and_symmetric_equals_generator(tmp_path, oracle, symmetric):
    """The `.mtx` input path on the unstructured config-5 input (sparse_matrix.hpp:225-357: threaded parse, symmetric
    expansion with the mirrored entry right behind its source entry, stable sort by row; utilities.hpp:326-367): unstr:20,20,20
    (24,000 rows, 1.6e6 entries) written as a general file and as a symmetric-lower, column-major file -- the layout of the
    SuiteSparse files config 5 names -- gives, through the CLI's binary CRS cache, the generator's CRS bit for bit, and the
    cache reload reproduces the residual table digit for digit.  (The same at 1.04e8 entries with the timings of every phase:
    tools/mtx_ingest.py, profiles/r05_d_mtx_ingest_unstr80.json.)"""
    import ctypes as C
    A = oracle.gen_unstr(20, 20, 20)
    oracle.lib.orc_write_mtx.restype = C.c_int64
    mtx, cache = str(tmp_path / "u.mtx"), str(tmp_path / "u.crs")
    stored = oracle.lib.orc_write_mtx(mtx.encode(), C.c_int64(A.n_rows), A.row_ptr.ctypes, A.col.ctypes, A.val.ctypes, C.c_int(symmetric))
    assert stored == (A.nnz if not symmetric else (A.nnz + A.n_rows) // 2)

    def run():
        out = subprocess.run([BIN, mtx, "-cg", "-p", "j", "-cache", cache], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("Matrix input:")]
        assert len(line) == 1
        return [float(v) for _, v in RES.findall(out.stdout)], line[0]

    first, l1 = run()
    assert ".mtx read + parse" in l1 and "COO -> CRS" in l1 and "upload" in l1
    with open(cache, "rb") as f:
        h = np.fromfile(f, dtype=np.int64, count=6)
        rp = np.fromfile(f, dtype=np.int64, count=int(h[1]) + 1)
        col = np.fromfile(f, dtype=np.int32, count=int(h[3]))
        val = np.fromfile(f, dtype=np.float64, count=int(h[3]))
    assert np.array_equal(rp, A.row_ptr) and np.array_equal(col, A.col) and np.array_equal(val, A.val)
    second, l2 = run()
    assert "binary CRS cache read" in l2 and second == first and len(first) > 10
    gen = subprocess.run([BIN, "unstr:20,20,20", "-cg", "-p", "j"], capture_output=True, text=True, timeout=600)
    assert [float(v) for _, v in RES.findall(gen.stdout)] == first
