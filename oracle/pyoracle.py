"""ctypes bindings for the CPU oracle and the compiled reference.

TEST INFRASTRUCTURE ONLY.  `Oracle` wraps oracle/libbisoracle.so (the C
restatement, oracle/bis_oracle.c); `Ref` wraps oracle/_ref/libbisref*.so (the
real reference compiled from /root/reference by oracle/Makefile).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

PC = dict(none=0, j=1, gs=2, bgs=3, sgs=4, **{"2st": 5, "s2st": 6, "ilu0": 7})
SOLVER = dict(j=0, gs=1, sgs=2, gm=3, cg=4, bi=5)

f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def set_omp_threads(n):
    """Set the OpenMP team size of the (single) libgomp the oracle and the
    compiled reference share -- OMP_NUM_THREADS is read only at load time."""
    C.CDLL("libgomp.so.1").omp_set_num_threads(C.c_int(int(n)))


def ref_build_info():
    """How and where oracle/_ref was compiled (written by oracle/Makefile next to the libraries)."""
    import json
    try:
        return json.load(open(os.path.join(HERE, "_ref", "build_info.json")))
    except Exception:
        return {}


def cpu_has_avx512():
    """The prebuilt libraries are compiled for x86-64-v4 (oracle/Makefile MARCH): a host without AVX-512 would die
    with SIGILL inside ctypes instead of failing a test."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return "avx512f" in line.split() and "avx512vl" in line.split()
    except OSError:
        pass
    return False


def host_march():
    return "x86-64-v4" if cpu_has_avx512() else "x86-64-v3"


def build(ref=True):
    """Compile the oracle (and, when /root/reference exists, oracle/_ref) for the ISA level this host has."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle", f"MARCH={host_march()}"])
    with open(os.path.join(HERE, ".oracle_march"), "w") as f:
        f.write(host_march())
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref", f"MARCH={host_march()}"])


def _oracle_usable():
    """libbisoracle.so exists and was built for an ISA level this CPU has (the file beside it says which; a library
    of unknown level counts as x86-64-v4, the Makefile's default)."""
    so = os.path.join(HERE, "libbisoracle.so")
    if not os.path.exists(so):
        return False
    try:
        built = open(os.path.join(HERE, ".oracle_march")).read().strip()
    except OSError:
        built = "x86-64-v4"
    return built != "x86-64-v4" or cpu_has_avx512()


class CRS:
    """Host CRS with int64 row_ptr / int32 col / float64 val."""

    def __init__(self, n_rows, row_ptr, col, val, n_cols=None):
        self.n_rows = int(n_rows)
        self.n_cols = int(n_cols if n_cols is not None else n_rows)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float64)
        self.nnz = int(self.row_ptr[-1]) if len(self.row_ptr) else 0

    @property
    def rp32(self):
        return self.row_ptr.astype(np.int32)

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val, self.col, self.row_ptr),
                             shape=(self.n_rows, self.n_cols))


class _OrcCrs(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_cols", C.c_int64),
                ("nnz", C.c_int64), ("row_ptr", C.c_void_p),
                ("col", C.c_void_p), ("val", C.c_void_p)]


class OrcOpts(C.Structure):
    _fields_ = [("solver", C.c_int), ("precond", C.c_int),
                ("max_iters", C.c_int), ("tol", C.c_double),
                ("restart_len", C.c_int), ("outer_iters", C.c_int),
                ("inner_iters", C.c_int), ("init_x", C.c_double),
                ("b_val", C.c_double), ("num_scale", C.c_int),
                ("ilu_pivot_tol", C.c_double), ("ilu_pivot_repl", C.c_double),
                ("ilu_real", C.c_int)]


class OrcResult(C.Structure):
    _fields_ = [("iters", C.c_int), ("n_hist", C.c_int),
                ("converged", C.c_int), ("stopping_criteria", C.c_double),
                ("final_true_residual", C.c_double)]


def _crs_struct(A):
    s = _OrcCrs(A.n_rows, A.n_cols, A.nnz, A.row_ptr.ctypes.data,
                A.col.ctypes.data, A.val.ctypes.data)
    s._keep = A
    return s


class Oracle:
    def __init__(self, path=None):
        if path is None:
            path = os.path.join(HERE, "libbisoracle.so")
            if not _oracle_usable():  # missing, or built for AVX-512 on a host without it: rebuild (gcc is enough)
                if os.path.exists(path):
                    os.remove(path)
                build(ref=False)
        self.lib = L = C.CDLL(path)
        L.orc_dot.restype = C.c_double
        L.orc_euclidean_vec_norm.restype = C.c_double
        L.orc_hpcg_row_ptr.restype = C.c_int64
        L.orc_anderson_diag.restype = C.c_double
        L.orc_peel_diag_crs.restype = C.c_int64
        L.orc_extract_scale.restype = C.c_int64

    def num_threads(self):
        return self.lib.orc_num_threads()

    # ---- kernels ---------------------------------------------------------
    def spmv(self, A, x):
        y = np.empty(A.n_rows)
        self.lib.orc_spmv(C.c_int64(A.n_rows), A.row_ptr.ctypes, A.col.ctypes,
                          A.val.ctypes, np.ascontiguousarray(x).ctypes, y.ctypes)
        return y

    def sptrsv(self, Ls, D, b, x=None, backward=False):
        """x may be the same array as b (in-place solve)."""
        if x is None:
            x = np.zeros(Ls.n_rows)
        fn = self.lib.orc_bsptrsv if backward else self.lib.orc_sptrsv
        fn(C.c_int64(Ls.n_rows), Ls.row_ptr.ctypes, Ls.col.ctypes,
           Ls.val.ctypes, x.ctypes, D.ctypes, b.ctypes)
        return x

    def _axpy(self, name, a, b, s, out=None):
        out = np.empty_like(a) if out is None else out
        getattr(self.lib, name)(out.ctypes, a.ctypes, b.ctypes,
                                C.c_int64(a.size), C.c_double(s))
        return out

    def subtract_vectors(self, a, b, s=1.0, out=None):
        return self._axpy("orc_subtract_vectors", a, b, s, out)

    def sum_vectors(self, a, b, s=1.0, out=None):
        return self._axpy("orc_sum_vectors", a, b, s, out)

    def elemwise_mult_vectors(self, a, b, s=1.0, out=None):
        return self._axpy("orc_elemwise_mult_vectors", a, b, s, out)

    def elemwise_div_vectors(self, a, b, s=1.0, out=None):
        return self._axpy("orc_elemwise_div_vectors", a, b, s, out)

    def dot(self, a, b):
        return self.lib.orc_dot(a.ctypes, b.ctypes, C.c_int64(a.size))

    def norm(self, v):
        return self.lib.orc_euclidean_vec_norm(v.ctypes, C.c_int64(v.size))

    def scale(self, v, s):
        out = np.empty_like(v)
        self.lib.orc_scale(out.ctypes, v.ctypes, C.c_double(s), C.c_int64(v.size))
        return out

    def normalize_x(self, x_new, x_old, D, b):
        x_new = x_new.copy()
        self.lib.orc_normalize_x(x_new.ctypes, x_old.ctypes, D.ctypes,
                                 b.ctypes, C.c_int64(x_new.size))
        return x_new

    def multi_axpy(self, V, y, n_vec):
        """V: (k, N) row-major; returns sum_{j<n_vec} y[j] V[j]."""
        N = V.shape[1]
        out = np.empty(N)
        y = np.ascontiguousarray(y, dtype=np.float64)
        self.lib.orc_multi_axpy(V.ctypes, C.c_int64(N), y.ctypes,
                                C.c_int(n_vec), out.ctypes)
        return out

    def compute_residual(self, A, x, b):
        res = np.empty(A.n_rows)
        tmp = np.empty(A.n_rows)
        s = _crs_struct(A)
        self.lib.orc_compute_residual(C.byref(s), x.ctypes, b.ctypes,
                                      res.ctypes, tmp.ctypes)
        return res

    def apply_preconditioner(self, pc, Ls, Us, A_D, A_D_inv, L_D, U_D, vin,
                             inplace=False, outer=1, inner=0):
        n = vin.size
        vin = vin.copy()
        out = vin if inplace else np.zeros(n)
        tmp = np.zeros(n)
        work = np.zeros(n)
        ls = _crs_struct(Ls) if Ls is not None else None
        us = _crs_struct(Us) if Us is not None else None
        one = np.ones(n)
        self.lib.orc_apply_preconditioner(
            C.c_int(PC[pc] if isinstance(pc, str) else pc), C.c_int64(n),
            C.byref(ls) if ls else None, C.byref(us) if us else None,
            (A_D if A_D is not None else one).ctypes,
            (A_D_inv if A_D_inv is not None else one).ctypes,
            (L_D if L_D is not None else one).ctypes,
            (U_D if U_D is not None else one).ctypes,
            out.ctypes, vin.ctypes, tmp.ctypes, work.ctypes,
            C.c_int(outer), C.c_int(inner))
        return out

    # ---- setup -----------------------------------------------------------
    def read_mtx(self, path):
        n_rows, n_cols, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        rp, col, val = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rc = self.lib.orc_read_mtx_crs(path.encode(), C.byref(n_rows),
                                       C.byref(n_cols), C.byref(nnz),
                                       C.byref(rp), C.byref(col), C.byref(val))
        if rc != 0:
            raise RuntimeError(f"orc_read_mtx_crs({path}) failed: {rc}")
        n, z = n_rows.value, nnz.value
        row_ptr = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_int64)), (n + 1,)).copy()
        c = np.ctypeslib.as_array(C.cast(col, C.POINTER(C.c_int32)), (max(z, 1),))[:z].copy()
        v = np.ctypeslib.as_array(C.cast(val, C.POINTER(C.c_double)), (max(z, 1),))[:z].copy()
        for p in (rp, col, val):
            self.lib.orc_free(p)
        return CRS(n, row_ptr, c, v)

    def coo_to_crs(self, n_rows, I, J, V):
        nnz = len(I)
        rp = np.zeros(n_rows + 1, dtype=np.int64)
        col = np.zeros(nnz, dtype=np.int32)
        val = np.zeros(nnz)
        rc = self.lib.orc_coo_to_crs(
            C.c_int64(n_rows), C.c_int64(nnz),
            np.ascontiguousarray(I, dtype=np.int32).ctypes,
            np.ascontiguousarray(J, dtype=np.int32).ctypes,
            np.ascontiguousarray(V, dtype=np.float64).ctypes,
            rp.ctypes, col.ctypes, val.ctypes)
        assert rc == 0
        return CRS(n_rows, rp, col, val)

    def split_LU(self, A):
        s = _crs_struct(A)
        c4 = np.zeros(4, dtype=np.int64)
        self.lib.orc_split_LU_count(C.byref(s), c4.ctypes)
        n = A.n_rows
        rps = [np.zeros(n + 1, dtype=np.int64) for _ in range(4)]
        cols = [np.zeros(int(c), dtype=np.int32) for c in c4]
        vals = [np.zeros(int(c)) for c in c4]
        args = []
        for k in range(4):
            args += [rps[k].ctypes, cols[k].ctypes, vals[k].ctypes]
        self.lib.orc_split_LU_fill(C.byref(s), *args)
        return [CRS(n, rps[k], cols[k], vals[k]) for k in range(4)]  # L, Ls, U, Us

    def peel_diag(self, M, want_inv=True):
        """In place on M; returns (D, D_inv, status)."""
        D = np.ones(M.n_rows)
        Di = np.zeros(M.n_rows)
        st = self.lib.orc_peel_diag_crs(C.c_int64(M.n_rows), M.row_ptr.ctypes,
                                        M.col.ctypes, M.val.ctypes, D.ctypes,
                                        Di.ctypes if want_inv else None)
        return D, Di, st

    def extract_scale(self, A):
        s = _crs_struct(A)
        out = np.zeros(A.n_rows)
        st = self.lib.orc_extract_scale(C.byref(s), out.ctypes)
        return out, st

    def scale_mat(self, A, s):
        self.lib.orc_scale_mat(C.c_int64(A.n_rows), A.row_ptr.ctypes,
                               A.col.ctypes, A.val.ctypes, s.ctypes)

    def factor_ilu0(self, A, pivot_tol=1e-8, pivot_repl=1e-4):
        s = _crs_struct(A)
        c4 = np.zeros(4, dtype=np.int64)
        self.lib.orc_split_LU_count(C.byref(s), c4.ctypes)
        n = A.n_rows
        Ls = CRS(n, np.zeros(n + 1, dtype=np.int64),
                 np.zeros(int(c4[1]), dtype=np.int32), np.zeros(int(c4[1])))
        Us = CRS(n, np.zeros(n + 1, dtype=np.int64),
                 np.zeros(int(c4[3]), dtype=np.int32), np.zeros(int(c4[3])))
        L_D = np.zeros(n)
        U_D = np.zeros(n)
        self.lib.orc_factor_ilu0(C.byref(s), C.c_double(pivot_tol),
                                 C.c_double(pivot_repl), Ls.row_ptr.ctypes,
                                 Ls.col.ctypes, Ls.val.ctypes, L_D.ctypes,
                                 Us.row_ptr.ctypes, Us.col.ctypes,
                                 Us.val.ctypes, U_D.ctypes)
        Ls.nnz = int(Ls.row_ptr[-1])
        Us.nnz = int(Us.row_ptr[-1])
        return Ls, L_D, Us, U_D

    # ---- generators ------------------------------------------------------
    def hpcg_row_ptr(self, row, nx, ny, nz):
        return self.lib.orc_hpcg_row_ptr(C.c_int64(row), C.c_int64(nx),
                                         C.c_int64(ny), C.c_int64(nz))

    def gen_hpcg(self, nx, ny=None, nz=None, row0=0, row1=None):
        ny = nx if ny is None else ny
        nz = nx if nz is None else nz
        N = nx * ny * nz
        row1 = N if row1 is None else row1
        nnz = self.hpcg_row_ptr(row1, nx, ny, nz) - self.hpcg_row_ptr(row0, nx, ny, nz)
        rp = np.zeros(row1 - row0 + 1, dtype=np.int64)
        col = np.zeros(nnz, dtype=np.int32)
        val = np.zeros(nnz)
        self.lib.orc_gen_hpcg(C.c_int64(nx), C.c_int64(ny), C.c_int64(nz),
                              C.c_int64(row0), C.c_int64(row1), rp.ctypes,
                              col.ctypes, val.ctypes)
        return CRS(row1 - row0, rp, col, val, n_cols=N)

    def gen_anderson(self, L, t=1.0, W=5.0, shift=0.0, seed=1, row0=0, row1=None):
        N = L ** 3
        row1 = N if row1 is None else row1
        n = row1 - row0
        rp = np.zeros(n + 1, dtype=np.int64)
        col = np.zeros(7 * n, dtype=np.int32)
        val = np.zeros(7 * n)
        self.lib.orc_gen_anderson(C.c_int64(L), C.c_double(t), C.c_double(W),
                                  C.c_double(shift), C.c_uint64(seed),
                                  C.c_int64(row0), C.c_int64(row1), rp.ctypes,
                                  col.ctypes, val.ctypes)
        return CRS(n, rp, col, val, n_cols=N)

    def gen_fem(self, nx, ny=None, nz=None, keep=85, seed=1, row0=0, row1=None):
        ny = nx if ny is None else ny
        nz = nx if nz is None else nz
        N = 3 * nx * ny * nz
        row1 = N if row1 is None else row1
        n = row1 - row0
        rp = np.zeros(n + 1, dtype=np.int64)
        self.lib.orc_gen_fem.restype = C.c_int64
        args = (C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_int(keep), C.c_uint64(seed),
                C.c_int64(row0), C.c_int64(row1))
        nnz = self.lib.orc_gen_fem(*args, rp.ctypes, None, None)
        col = np.zeros(nnz, dtype=np.int32)
        val = np.zeros(nnz)
        self.lib.orc_gen_fem(*args, rp.ctypes, col.ctypes, val.ctypes)
        return CRS(n, rp, col, val, n_cols=N)

    def unstr_perm(self, n, seed=1):
        perm = np.empty(n, dtype=np.int32)
        if self.lib.orc_unstr_perm(C.c_int64(n), C.c_uint64(seed), perm.ctypes):
            raise MemoryError
        return perm

    def gen_unstr(self, nx, ny=None, nz=None, keep=85, seed=1):
        """The FEM-like matrix under a seeded random symmetric row permutation, ascending columns (orc_gen_unstr)."""
        ny = nx if ny is None else ny
        nz = nx if nz is None else nz
        n = 3 * nx * ny * nz
        perm = self.unstr_perm(n, seed)
        rp = np.zeros(n + 1, dtype=np.int64)
        self.lib.orc_gen_unstr.restype = C.c_int64
        args = (C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_int(keep), C.c_uint64(seed), perm.ctypes)
        nnz = self.lib.orc_gen_unstr(*args, rp.ctypes, None, None)
        col = np.empty(nnz, dtype=np.int32)
        val = np.empty(nnz)
        self.lib.orc_gen_unstr(*args, rp.ctypes, col.ctypes, val.ctypes)
        return CRS(n, rp, col, val)

    def cg_run(self, A, iters, A_D=None, b_val=1.0, x0_val=0.1):
        """Plain CG loop for bench.py's cpu_baseline; returns (hist, seconds)."""
        hist = np.zeros(iters + 1)
        secs = C.c_double()
        self.lib.orc_cg_run(C.c_int64(A.n_rows), A.row_ptr.ctypes, A.col.ctypes,
                            A.val.ctypes, A_D.ctypes if A_D is not None else None,
                            C.c_double(b_val), C.c_double(x0_val), C.c_int(iters),
                            hist.ctypes, C.byref(secs))
        return hist, secs.value

    # ---- full solve --------------------------------------------------------
    def solve(self, A, solver, precond="none", max_iters=1000, tol=1e-14,
              restart_len=10, outer=1, inner=0, init_x=0.1, b_val=1.0,
              num_scale=False, ilu_real=False):
        o = OrcOpts(SOLVER[solver], PC[precond], max_iters, tol, restart_len,
                    outer, inner, init_x, b_val, int(num_scale), 1e-8, 1e-4,
                    int(ilu_real))
        hist = np.zeros(2 * max_iters)
        x_star = np.zeros(A.n_rows)
        res = OrcResult()
        rc = self.lib.orc_solve(C.c_int64(A.n_rows), C.c_int64(A.nnz),
                                A.row_ptr.ctypes, A.col.ctypes, A.val.ctypes,
                                C.byref(o), hist.ctypes, x_star.ctypes,
                                C.byref(res))
        if rc != 0:
            raise RuntimeError(f"orc_solve failed: {rc}")
        return dict(iters=res.iters, hist=hist[:res.n_hist].copy(),
                    converged=bool(res.converged),
                    stopping=res.stopping_criteria,
                    final_true_residual=res.final_true_residual, x=x_star)


class Ref:
    """The compiled reference (oracle/_ref).  Present only where oracle/_ref
    was built (this container) or shipped prebuilt (the GPU box)."""

    def __init__(self, variant=""):
        path = os.path.join(HERE, "_ref", f"libbisref{variant}.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        if not Ref.available(variant):
            raise RuntimeError(f"{path} was built for {ref_build_info().get('flags')} and this CPU lacks AVX-512")
        self.lib = L = C.CDLL(path)
        L.ref_dot.restype = C.c_double
        L.ref_euclidean_vec_norm.restype = C.c_double

    @staticmethod
    def available(variant=""):
        """The prebuilt reference can be loaded here: the file exists and this CPU has the ISA level it was
        compiled for (build_info.json; it cannot be rebuilt where /root/reference is absent)."""
        if not os.path.exists(os.path.join(HERE, "_ref", f"libbisref{variant}.so")):
            return False
        return "x86-64-v4" not in ref_build_info().get("flags", "x86-64-v4") or cpu_has_avx512()

    def set_first_touch(self, on):
        """1: the solver's matrix copy is made by the reference's own MatrixCRS::operator= (an OpenMP-parallel copy:
        rows first touched by the threads that multiply them, sparse_matrix.hpp:92-128) instead of a memcpy by the
        calling thread."""
        self.lib.ref_set_first_touch(C.c_int(int(bool(on))))

    def _m(self, A):
        return (C.c_int(A.n_rows), C.c_int(A.nnz), A.rp32.ctypes, A.col.ctypes,
                A.val.ctypes)

    def spmv(self, A, x):
        y = np.empty(A.n_rows)
        rp = A.rp32
        self.lib.ref_spmv(C.c_int(A.n_rows), C.c_int(A.nnz), rp.ctypes,
                          A.col.ctypes, A.val.ctypes, x.ctypes, y.ctypes)
        return y

    def sptrsv(self, Ls, D, b, x=None, backward=False):
        if x is None:
            x = np.zeros(Ls.n_rows)
        rp = Ls.rp32
        fn = self.lib.ref_bsptrsv if backward else self.lib.ref_sptrsv
        fn(C.c_int(Ls.n_rows), C.c_int(Ls.nnz), rp.ctypes, Ls.col.ctypes,
           Ls.val.ctypes, x.ctypes, D.ctypes, b.ctypes)
        return x

    def _axpy(self, name, a, b, s):
        out = np.empty_like(a)
        getattr(self.lib, name)(out.ctypes, a.ctypes, b.ctypes,
                                C.c_int(a.size), C.c_double(s))
        return out

    def subtract_vectors(self, a, b, s=1.0):
        return self._axpy("ref_subtract_vectors", a, b, s)

    def sum_vectors(self, a, b, s=1.0):
        return self._axpy("ref_sum_vectors", a, b, s)

    def elemwise_mult_vectors(self, a, b, s=1.0):
        return self._axpy("ref_elemwise_mult_vectors", a, b, s)

    def elemwise_div_vectors(self, a, b, s=1.0):
        return self._axpy("ref_elemwise_div_vectors", a, b, s)

    def dot(self, a, b):
        return self.lib.ref_dot(a.ctypes, b.ctypes, C.c_int(a.size))

    def norm(self, v):
        return self.lib.ref_euclidean_vec_norm(v.ctypes, C.c_int(v.size))

    def scale(self, v, s):
        out = np.empty_like(v)
        self.lib.ref_scale(out.ctypes, v.ctypes, C.c_double(s), C.c_int(v.size))
        return out

    def normalize_x(self, x_new, x_old, D, b):
        x_new = x_new.copy()
        self.lib.ref_normalize_x(x_new.ctypes, x_old.ctypes, D.ctypes,
                                 b.ctypes, C.c_int(x_new.size))
        return x_new

    def dgemm_transpose1(self, V, y, n_vec):
        N = V.shape[1]
        out = np.empty(N)
        y = np.ascontiguousarray(y, dtype=np.float64)
        self.lib.ref_dgemm_transpose1(V.ctypes, y.ctypes, out.ctypes,
                                      C.c_int(N), C.c_int(n_vec))
        return out

    def compute_residual(self, A, x, b):
        res = np.empty(A.n_rows)
        tmp = np.empty(A.n_rows)
        rp = A.rp32
        self.lib.ref_compute_residual(C.c_int(A.n_rows), C.c_int(A.nnz),
                                      rp.ctypes, A.col.ctypes, A.val.ctypes,
                                      x.ctypes, b.ctypes, res.ctypes, tmp.ctypes)
        return res

    def apply_preconditioner(self, pc, Ls, Us, A_D, A_D_inv, L_D, U_D, vin,
                             inplace=False):
        n = vin.size
        vin = vin.copy()
        out = vin if inplace else np.zeros(n)
        tmp = np.zeros(n)
        work = np.zeros(n)
        one = np.ones(n)
        empty = CRS(n, np.zeros(n + 1, dtype=np.int64), np.zeros(0, np.int32),
                    np.zeros(0))
        Ls = Ls or empty
        Us = Us or empty
        lrp, urp = Ls.rp32, Us.rp32
        self.lib.ref_apply_preconditioner(
            C.c_int(PC[pc] if isinstance(pc, str) else pc), C.c_int(n),
            C.c_int(Ls.nnz), lrp.ctypes, Ls.col.ctypes, Ls.val.ctypes,
            C.c_int(Us.nnz), urp.ctypes, Us.col.ctypes, Us.val.ctypes,
            (A_D if A_D is not None else one).ctypes,
            (A_D_inv if A_D_inv is not None else one).ctypes,
            (L_D if L_D is not None else one).ctypes,
            (U_D if U_D is not None else one).ctypes,
            out.ctypes, vin.ctypes, tmp.ctypes, work.ctypes)
        return out

    def read_mtx(self, path):
        n, nnz = C.c_int(), C.c_int()
        rc = self.lib.ref_read_mtx(path.encode(), C.byref(n), C.byref(nnz))
        if rc != 0:
            raise RuntimeError("ref_read_mtx failed")
        rp = np.zeros(n.value + 1, dtype=np.int32)
        col = np.zeros(nnz.value, dtype=np.int32)
        val = np.zeros(nnz.value)
        self.lib.ref_read_mtx_fetch(rp.ctypes, col.ctypes, val.ctypes)
        return CRS(n.value, rp, col, val)

    def coo_to_crs(self, n_rows, I, J, V):
        nnz = len(I)
        rp = np.zeros(n_rows + 1, dtype=np.int32)
        col = np.zeros(nnz, dtype=np.int32)
        val = np.zeros(nnz)
        self.lib.ref_coo_to_crs(C.c_int(n_rows), C.c_int(nnz),
                                np.ascontiguousarray(I, dtype=np.int32).ctypes,
                                np.ascontiguousarray(J, dtype=np.int32).ctypes,
                                np.ascontiguousarray(V, dtype=np.float64).ctypes,
                                rp.ctypes, col.ctypes, val.ctypes)
        return CRS(n_rows, rp, col, val)

    def split_LU(self, A):
        c4 = np.zeros(4, dtype=np.int32)
        rp = A.rp32
        self.lib.ref_split_LU(C.c_int(A.n_rows), C.c_int(A.nnz), rp.ctypes,
                              A.col.ctypes, A.val.ctypes, c4.ctypes)
        out = []
        for k in range(4):
            r = np.zeros(A.n_rows + 1, dtype=np.int32)
            c = np.zeros(int(c4[k]), dtype=np.int32)
            v = np.zeros(int(c4[k]))
            self.lib.ref_split_LU_fetch(C.c_int(k), r.ctypes, c.ctypes, v.ctypes)
            out.append(CRS(A.n_rows, r, c, v))
        return out

    def peel_diag(self, M):
        D = np.ones(M.n_rows)
        Di = np.zeros(M.n_rows)
        rp = M.rp32
        self.lib.ref_peel_diag_crs(C.c_int(M.n_rows), C.c_int(M.nnz), rp.ctypes,
                                   M.col.ctypes, M.val.ctypes, D.ctypes, Di.ctypes)
        return D, Di

    def extract_scale(self, A):
        out = np.zeros(A.n_rows)
        rp = A.rp32
        self.lib.ref_extract_scale(C.c_int(A.n_rows), C.c_int(A.nnz), rp.ctypes,
                                   A.col.ctypes, A.val.ctypes, out.ctypes)
        return out

    def factor_ilu0(self, A):
        n = A.n_rows
        rp = A.rp32
        Lrp = np.zeros(n + 1, dtype=np.int32)
        Urp = np.zeros(n + 1, dtype=np.int32)
        Lc = np.zeros(A.nnz, dtype=np.int32)
        Uc = np.zeros(A.nnz, dtype=np.int32)
        Lv = np.zeros(A.nnz)
        Uv = np.zeros(A.nnz)
        L_D = np.zeros(n)
        U_D = np.zeros(n)
        nnz2 = np.zeros(2, dtype=np.int32)
        self.lib.ref_factor_ilu0(C.c_int(n), C.c_int(A.nnz), rp.ctypes,
                                 A.col.ctypes, A.val.ctypes, Lrp.ctypes,
                                 Lc.ctypes, Lv.ctypes, L_D.ctypes, Urp.ctypes,
                                 Uc.ctypes, Uv.ctypes, U_D.ctypes, nnz2.ctypes)
        Ls = CRS(n, Lrp, Lc[:nnz2[0]], Lv[:nnz2[0]])
        Us = CRS(n, Urp, Uc[:nnz2[1]], Uv[:nnz2[1]])
        return Ls, L_D, Us, U_D

    def solve(self, A, solver, precond="none", max_iters=1000, tol=1e-14,
              restart_len=10, num_scale=False, ilu_real=False):
        hist = np.zeros(2 * 1000)
        x_star = np.zeros(A.n_rows)
        oi = np.zeros(3, dtype=np.int32)
        od = np.zeros(5)
        rp = A.rp32
        rc = self.lib.ref_solve(C.c_int(A.n_rows), C.c_int(A.nnz), rp.ctypes,
                                A.col.ctypes, A.val.ctypes,
                                C.c_int(SOLVER[solver]), C.c_int(PC[precond]),
                                C.c_int(restart_len), C.c_int(int(num_scale)),
                                C.c_int(max_iters), C.c_double(tol),
                                C.c_int(int(ilu_real)), hist.ctypes,
                                x_star.ctypes, oi.ctypes, od.ctypes)
        if rc != 0:
            raise RuntimeError(f"ref_solve failed: {rc}")
        return dict(iters=int(oi[0]), hist=hist[:oi[1]].copy(),
                    converged=bool(oi[2]), stopping=od[0],
                    final_true_residual=od[1], x=x_star, iterate_s=od[2],
                    sample_s=od[3], spmv_s=od[4])
