"""basic_iterative_solvers_amd -- MI355X (gfx950) implementation of the SpMV +
preconditioner-apply + BLAS-1 hot path of DanecLacey/basic_iterative_solvers.

The product is the C-ABI library `lib/libbis_hip.so` (include/bis_hip.h),
hand-written HIP for gfx950, plus the C++ host layer under `host/` that mirrors
the reference's operator surface.  This Python module is only a thin ctypes
binding used by the tests, `bench.py` and `__graft_entry__.py`; it adds no
compute path of its own and there is no CPU fallback: without a usable gfx950
device `Context()` raises.
"""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "lib", "libbis_hip.so")

PC = dict(none=0, j=1, gs=2, bgs=3, sgs=4, **{"2st": 5, "s2st": 6, "ilu0": 7})

_lib = None


class BisError(RuntimeError):
    pass


def load_library():
    """dlopen the in-tree C-ABI library (never builds, never falls back)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BisError(
                f"{LIB_PATH} is missing: build it with "
                "`python -m basic_iterative_solvers_amd.build` (hipcc, gfx950)")
        _lib = C.CDLL(LIB_PATH)
        _lib.bis_last_error.restype = C.c_char_p
        _lib.bis_ctx_stream.restype = C.c_void_p
        _lib.bis_mat_sweep_kernel.restype = C.c_char_p
    return _lib


def _i64(v):
    return C.c_int64(int(v))


class Context:
    """bis_ctx: one device + stream.  Raises if no gfx950 device is usable."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        self.h = C.c_void_p()
        st = self.lib.bis_ctx_create(C.c_int(device), C.c_void_p(stream), C.byref(self.h))
        if st != 0:
            raise BisError(f"bis_ctx_create failed (status {st}): no usable gfx950 device; "
                           "this package has no CPU fallback")

    def check(self, st):
        if st != 0:
            raise BisError(f"status {st}: {self.lib.bis_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.lib.bis_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def sync(self):
        self.check(self.lib.bis_sync(self.h))

    def set_option(self, name, value):
        if self.lib.bis_set_option(name.encode(), C.c_int(int(value))) != 0:
            raise BisError(f"bis_set_option: unknown option {name}")

    def options(self):
        """The options in effect (bis_options_describe): every option not at its default + the BIS_* variables found in the
        environment at first use -- for bench / test records."""
        import json
        n = self.lib.bis_options_describe(None, C.c_int(0))
        buf = C.create_string_buffer(n + 1)
        self.lib.bis_options_describe(buf, C.c_int(n + 1))
        return json.loads(buf.value.decode())

    def device_info(self):
        arch = C.create_string_buffer(64)
        n_cus = C.c_int()
        hbm = C.c_int64()
        self.check(self.lib.bis_device_info(self.h, arch, C.c_size_t(64), C.byref(n_cus),
                                            C.byref(hbm)))
        return dict(arch=arch.value.decode(), n_cus=n_cus.value, hbm_bytes=hbm.value)

    # ---- vectors -----------------------------------------------------------
    def alloc(self, n):
        p = C.c_void_p()
        self.check(self.lib.bis_vec_alloc(self.h, _i64(n), C.byref(p)))
        return Vec(self, p.value, int(n), True)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        v = self.alloc(arr.size)
        self.check(self.lib.bis_vec_upload(self.h, C.c_void_p(v.ptr), arr.ctypes, _i64(arr.size)))
        return v

    # ---- matrices ----------------------------------------------------------
    def matrix(self, crs):
        """Upload a host CRS (object with n_rows, n_cols, row_ptr(int64), col, val)."""
        h = C.c_void_p()
        rp = np.ascontiguousarray(crs.row_ptr, dtype=np.int64)
        col = np.ascontiguousarray(crs.col, dtype=np.int32)
        val = np.ascontiguousarray(crs.val, dtype=np.float64)
        nnz = int(rp[-1]) if len(rp) else 0
        if nnz < 2 ** 31 - 16:
            rp32 = rp.astype(np.int32)
            st = self.lib.bis_mat_create(self.h, _i64(crs.n_rows), _i64(crs.n_cols), _i64(nnz),
                                         rp32.ctypes, col.ctypes, val.ctypes, C.byref(h))
        else:
            st = self.lib.bis_mat_create64(self.h, _i64(crs.n_rows), _i64(crs.n_cols), _i64(nnz),
                                           rp.ctypes, col.ctypes, val.ctypes, C.byref(h))
        self.check(st)
        return Mat(self, h)

    def gen_hpcg(self, nx, ny=None, nz=None, row0=0, row1=None):
        ny = nx if ny is None else ny
        nz = nx if nz is None else nz
        row1 = nx * ny * nz if row1 is None else row1
        h = C.c_void_p()
        self.check(self.lib.bis_mat_gen_hpcg(self.h, _i64(nx), _i64(ny), _i64(nz), _i64(row0),
                                             _i64(row1), C.byref(h)))
        return Mat(self, h)

    def gen_anderson(self, L, t=1.0, W=5.0, shift=0.0, seed=1, row0=0, row1=None):
        row1 = L ** 3 if row1 is None else row1
        h = C.c_void_p()
        self.check(self.lib.bis_mat_gen_anderson(self.h, _i64(L), C.c_double(t), C.c_double(W),
                                                 C.c_double(shift), C.c_uint64(seed),
                                                 _i64(row0), _i64(row1), C.byref(h)))
        return Mat(self, h)

    def gen_fem(self, nx, ny=None, nz=None, keep=85, seed=1, row0=0, row1=None):
        ny = nx if ny is None else ny
        nz = nx if nz is None else nz
        row1 = 3 * nx * ny * nz if row1 is None else row1
        h = C.c_void_p()
        self.check(self.lib.bis_mat_gen_fem(self.h, _i64(nx), _i64(ny), _i64(nz), C.c_int(keep),
                                            C.c_uint64(seed), _i64(row0), _i64(row1), C.byref(h)))
        return Mat(self, h)

    def gen_unstr(self, nx, ny=None, nz=None, keep=85, seed=1):
        """fem: under a seeded random symmetric row permutation, ascending columns, no grid hint (config 5's unstructured input)."""
        ny = nx if ny is None else ny
        nz = nx if nz is None else nz
        h = C.c_void_p()
        self.check(self.lib.bis_mat_gen_unstr(self.h, _i64(nx), _i64(ny), _i64(nz), C.c_int(keep), C.c_uint64(seed),
                                              None, C.byref(h)))
        return Mat(self, h)

    def multicolour(self, A):
        """B = P A P^T grouped by colour; returns (B, perm[new]=old as numpy int32, n_colours)."""
        n = A.n_rows
        store = self.alloc((n + 1) // 2 + 1)
        h, nc = C.c_void_p(), C.c_int()
        self.check(self.lib.bis_mat_multicolour(self.h, A.h, C.byref(h), C.c_void_p(store.ptr), C.byref(nc)))
        perm = store.to_host().view(np.int32)[:n].copy()
        store.free()
        return Mat(self, h), perm, nc.value

    def bfs_order(self, A, rcm=False):
        """Breadth-first / reverse Cuthill-McKee permutation on the device (perm[new] = old, numpy int32)."""
        n = A.n_rows
        store = self.alloc((n + 1) // 2 + 1)
        self.check(self.lib.bis_mat_bfs_order(self.h, A.h, C.c_int(int(rcm)), C.c_void_p(store.ptr)))
        perm = store.to_host().view(np.int32)[:n].copy()
        store.free()
        return perm

    def permute(self, A, perm):
        """B = P A P^T on the device for perm[new] = old."""
        n = len(perm)
        raw = np.zeros((n + 1) // 2 + 1)
        raw.view(np.int32)[:n] = np.asarray(perm, dtype=np.int32)
        store = self.upload(raw)
        h = C.c_void_p()
        try:
            self.check(self.lib.bis_mat_permute(self.h, A.h, C.c_void_p(store.ptr), C.byref(h)))
        finally:
            store.free()
        return Mat(self, h)

    def scale_sym(self, A, init=0.0):
        """-scale on the device: returns the scale vector s (Vec); A's values are scaled in place."""
        s = self.alloc(A.n_rows)
        self.init_vector(s, init)
        self.check(self.lib.bis_mat_scale_sym(self.h, A.h, C.c_void_p(s.ptr)))
        return s

    def gather(self, out, vec, perm):
        """out[i] = vec[perm[i]] on the device (perm: numpy int32, uploaded for the call)."""
        n = len(perm)
        raw = np.zeros((n + 1) // 2 + 1)
        raw.view(np.int32)[:n] = np.asarray(perm, dtype=np.int32)
        store = self.upload(raw)
        self.check(self.lib.bis_vec_gather(self.h, C.c_void_p(out.ptr), C.c_void_p(vec.ptr), C.c_void_p(store.ptr), _i64(n)))
        self.sync()
        store.free()

    def tune_placement(self, A, max_trials=6):
        """Keep the fastest of up to max_trials re-allocations of A's streamed arrays; returns (first_ms, best_ms)."""
        f, b = C.c_double(), C.c_double()
        self.check(self.lib.bis_mat_tune_placement(self.h, A.h, C.c_int(max_trials), C.byref(f), C.byref(b)))
        return f.value, b.value

    def split_strict(self, A):
        n = A.n_rows
        D, Dinv = self.alloc(n), self.alloc(n)
        hl, hu = C.c_void_p(), C.c_void_p()
        self.check(self.lib.bis_mat_split_strict(self.h, A.h, C.byref(hl), C.byref(hu),
                                                 C.c_void_p(D.ptr), C.c_void_p(Dinv.ptr)))
        return Mat(self, hl), Mat(self, hu), D, Dinv

    def mat_diag(self, A, row_offset=0):
        """(D, 1/D) of a row block with global column indices (bis_mat_diag)."""
        D, Dinv = self.alloc(A.n_rows), self.alloc(A.n_rows)
        self.check(self.lib.bis_mat_diag(self.h, A.h, _i64(row_offset), C.c_void_p(D.ptr), C.c_void_p(Dinv.ptr)))
        return D, Dinv

    def diag_block(self, A, row_offset=0):
        """Square diagonal block of a row block with global columns (bis_mat_diag_block)."""
        h = C.c_void_p()
        self.check(self.lib.bis_mat_diag_block(self.h, A.h, _i64(row_offset), C.byref(h)))
        return Mat(self, h)

    def ilu0(self, A, pivot_tol=1e-8, pivot_repl=1e-4):
        n = A.n_rows
        L_D, U_D = self.alloc(n), self.alloc(n)
        hl, hu = C.c_void_p(), C.c_void_p()
        self.check(self.lib.bis_mat_ilu0(self.h, A.h, C.c_double(pivot_tol), C.c_double(pivot_repl),
                                         C.byref(hl), C.byref(hu), C.c_void_p(L_D.ptr),
                                         C.c_void_p(U_D.ptr)))
        return Mat(self, hl), L_D, Mat(self, hu), U_D

    # ---- kernels (kernels.hpp names) ---------------------------------------
    def spmv(self, A, x, y):
        self.check(self.lib.bis_spmv(self.h, A.h, C.c_void_p(x.ptr), C.c_void_p(y.ptr)))

    def sptrsv(self, Ls, x, D, b):
        self.check(self.lib.bis_sptrsv(self.h, Ls.h, C.c_void_p(x.ptr), C.c_void_p(D.ptr),
                                       C.c_void_p(b.ptr)))

    def bsptrsv(self, Us, x, D, b):
        self.check(self.lib.bis_bsptrsv(self.h, Us.h, C.c_void_p(x.ptr), C.c_void_p(D.ptr),
                                        C.c_void_p(b.ptr)))

    def _ew3(self, fn, r, a, b, scale, n=None):
        n = r.n if n is None else n
        self.check(fn(self.h, C.c_void_p(r.ptr), C.c_void_p(a.ptr), C.c_void_p(b.ptr), _i64(n),
                      C.c_double(scale)))

    def subtract_vectors(self, r, a, b, scale=1.0, n=None):
        self._ew3(self.lib.bis_subtract_vectors, r, a, b, scale, n)

    def sum_vectors(self, r, a, b, scale=1.0, n=None):
        self._ew3(self.lib.bis_sum_vectors, r, a, b, scale, n)

    def elemwise_mult_vectors(self, r, a, b, scale=1.0, n=None):
        self._ew3(self.lib.bis_elemwise_mult_vectors, r, a, b, scale, n)

    def elemwise_div_vectors(self, r, a, b, scale=1.0, n=None):
        self._ew3(self.lib.bis_elemwise_div_vectors, r, a, b, scale, n)

    def compute_residual(self, A, x, b, res, tmp):
        self.check(self.lib.bis_compute_residual(self.h, A.h, C.c_void_p(x.ptr), C.c_void_p(b.ptr),
                                                 C.c_void_p(res.ptr), C.c_void_p(tmp.ptr)))

    def dot(self, a, b, n=None):
        out = C.c_double()
        self.check(self.lib.bis_dot(self.h, C.c_void_p(a.ptr), C.c_void_p(b.ptr),
                                    _i64(a.n if n is None else n), C.byref(out)))
        return out.value

    def euclidean_vec_norm(self, v, n=None):
        out = C.c_double()
        self.check(self.lib.bis_euclidean_vec_norm(self.h, C.c_void_p(v.ptr),
                                                   _i64(v.n if n is None else n), C.byref(out)))
        return out.value

    def scale(self, r, v, scalar, n=None):
        self.check(self.lib.bis_scale(self.h, C.c_void_p(r.ptr), C.c_void_p(v.ptr),
                                      C.c_double(scalar), _i64(r.n if n is None else n)))

    def init_vector(self, v, val, n=None):
        self.check(self.lib.bis_init_vector(self.h, C.c_void_p(v.ptr), C.c_double(val),
                                            _i64(v.n if n is None else n)))

    def copy_vector(self, out, inp, n=None):
        self.check(self.lib.bis_copy_vector(self.h, C.c_void_p(out.ptr), C.c_void_p(inp.ptr),
                                            _i64(out.n if n is None else n)))

    def normalize_x(self, x_new, x_old, D, b):
        self.check(self.lib.bis_normalize_x(self.h, C.c_void_p(x_new.ptr), C.c_void_p(x_old.ptr),
                                            C.c_void_p(D.ptr), C.c_void_p(b.ptr), _i64(x_new.n)))

    def multi_axpy(self, V, ldv, y_host, n_vec, out, n):
        y = np.ascontiguousarray(y_host, dtype=np.float64)
        self.check(self.lib.bis_multi_axpy(self.h, C.c_void_p(V.ptr), _i64(ldv), y.ctypes,
                                           C.c_int(n_vec), C.c_void_p(out.ptr), _i64(n)))

    def apply_preconditioner(self, pc, n, Ls, Us, A_D, A_D_inv, L_D, U_D, out, inp, tmp, work,
                             outer=1, inner=0):
        def p(v):
            return C.c_void_p(v.ptr) if v is not None else C.c_void_p()
        self.check(self.lib.bis_apply_preconditioner(
            self.h, C.c_int(PC[pc] if isinstance(pc, str) else pc), _i64(n),
            Ls.h if Ls is not None else C.c_void_p(), Us.h if Us is not None else C.c_void_p(),
            p(A_D), p(A_D_inv), p(L_D), p(U_D), p(out), p(inp), p(tmp), p(work),
            C.c_int(outer), C.c_int(inner)))

    # ---- fused CG ------------------------------------------------------------
    def cg(self, A, b, x, A_D=None):
        return CG(self, A, b, x, A_D)

    # ---- measurement ---------------------------------------------------------
    def stat(self, kind, A, D, b, x, Ls=None, Us=None):
        return Stat(self, kind, A, D, b, x, Ls, Us)

    def profile(self, on):
        self.check(self.lib.bis_profile_enable(self.h, C.c_int(int(on))))

    def profile_read(self):
        n = C.c_int64()
        ms = C.c_double()
        self.check(self.lib.bis_profile_read(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_read_sweeps(self):
        """(sweeps, summed ms) of the bis_sptrsv / bis_bsptrsv calls made while profiling was on (HIP events on the library's stream)."""
        n = C.c_int64()
        ms = C.c_double()
        self.check(self.lib.bis_profile_read_sweeps(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value


class Vec:
    """A device vector: raw `double*` + length (what the reference passes as
    `double *`).  `offset(k)` gives the pointer-arithmetic view `&v[k]`
    (gmres.hpp:169)."""

    def __init__(self, ctx, ptr, n, owner):
        self.ctx, self.ptr, self.n, self.owner = ctx, ptr, n, owner

    def offset(self, k, n=None):
        return Vec(self.ctx, self.ptr + 8 * int(k), self.n - int(k) if n is None else n, False)

    def to_host(self):
        out = np.empty(self.n, dtype=np.float64)
        self.ctx.check(self.ctx.lib.bis_vec_download(self.ctx.h, out.ctypes, C.c_void_p(self.ptr),
                                                     _i64(self.n)))
        return out

    def set(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        assert arr.size == self.n
        self.ctx.check(self.ctx.lib.bis_vec_upload(self.ctx.h, C.c_void_p(self.ptr), arr.ctypes,
                                                   _i64(self.n)))

    def free(self):
        if self.owner and self.ptr:
            self.ctx.lib.bis_vec_free(self.ctx.h, C.c_void_p(self.ptr))
            self.ptr = 0


class Mat:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h
        n_rows, n_cols, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        ctx.lib.bis_mat_info(h, C.byref(n_rows), C.byref(n_cols), C.byref(nnz))
        self.n_rows, self.n_cols, self.nnz = n_rows.value, n_cols.value, nnz.value

    @property
    def rp_width(self):
        return int(self.ctx.lib.bis_mat_rp_width(self.h))

    def spmv_stream_info(self):
        """(col_bytes, val_bytes, n_dict, form) of the SpMV's streams for this matrix (bis_mat_spmv_stream_info)."""
        c, v, d, f = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(self.ctx.lib.bis_mat_spmv_stream_info(self.ctx.h, self.h, C.byref(c), C.byref(v), C.byref(d), C.byref(f)))
        return c.value, v.value, d.value, f.value

    def spmv_streamed_bytes(self):
        """Bytes one SpMV launch moves at least with the matrix' current stream format (bis_mat_spmv_streamed_bytes)."""
        b = C.c_int64()
        self.ctx.check(self.ctx.lib.bis_mat_spmv_streamed_bytes(self.ctx.h, self.h, C.byref(b)))
        return b.value

    def debug_ptrs(self):
        """Device addresses (row_ptr, col, val) of the CRS arrays (bis_mat_debug_ptrs)."""
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.ctx.check(self.ctx.lib.bis_mat_debug_ptrs(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def win8_tuning(self):
        """(re-allocations tried, kernel ms on the first allocation, kernel ms on the one kept) of the placement search the library
        made when it built the matrix' window + sliced-ELL stream (bis_mat_win8_tuning); zeros without one."""
        t, a, b = C.c_int(), C.c_double(), C.c_double()
        self.ctx.lib.bis_mat_win8_tuning(self.h, C.byref(t), C.byref(a), C.byref(b))
        return t.value, a.value, b.value

    def colslab_info(self):
        """(K, one-pass ms, K-passes ms): the column slabs the SpMV of this matrix runs on (0: none) and the build-time trial's
        times (bis_mat_colslab_info)."""
        k, a, b = C.c_int(), C.c_double(), C.c_double()
        self.ctx.lib.bis_mat_colslab_info(self.h, C.byref(k), C.byref(a), C.byref(b))
        return k.value, a.value, b.value

    def sweep_kernel(self, backward=False):
        """Name of the kernel the last forward / backward sweep on this triangle ran (bis_mat_sweep_kernel)."""
        return self.ctx.lib.bis_mat_sweep_kernel(self.h, C.c_int(int(backward))).decode()

    def retune(self):
        """Rebuild everything derived from the CRS arrays (bis_mat_retune): required after writing values in place."""
        self.ctx.check(self.ctx.lib.bis_mat_retune(self.ctx.h, self.h))

    def set_grid_hint(self, nx, ny, nz, dof=1):
        self.ctx.check(self.ctx.lib.bis_mat_set_grid_hint(self.h, _i64(nx), _i64(ny), _i64(nz), C.c_int(dof)))

    def download(self):
        rp = np.zeros(self.n_rows + 1, dtype=np.int64)
        col = np.zeros(self.nnz, dtype=np.int32)
        val = np.zeros(self.nnz, dtype=np.float64)
        self.ctx.check(self.ctx.lib.bis_mat_download(self.ctx.h, self.h, rp.ctypes, col.ctypes,
                                                     val.ctypes))
        return rp, col, val

    def free(self):
        if self.h:
            self.ctx.lib.bis_mat_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()


class CG:
    def __init__(self, ctx, A, b, x, A_D=None):
        self.ctx = ctx
        self.h = C.c_void_p()
        ctx.check(ctx.lib.bis_cg_create(ctx.h, A.h, C.c_void_p(A_D.ptr) if A_D else C.c_void_p(),
                                        C.c_void_p(b.ptr), C.c_void_p(x.ptr), C.byref(self.h)))

    def set_preconditioner(self, pc, Ls=None, Us=None, A_D=None, A_D_inv=None, L_D=None, U_D=None, outer=1, inner=0):
        def p(v):
            return C.c_void_p(v.ptr) if v is not None else C.c_void_p()
        self._keep = (Ls, Us, A_D, A_D_inv, L_D, U_D)
        self.ctx.check(self.ctx.lib.bis_cg_set_preconditioner(
            self.ctx.h, self.h, C.c_int(PC[pc] if isinstance(pc, str) else pc),
            Ls.h if Ls is not None else C.c_void_p(), Us.h if Us is not None else C.c_void_p(),
            p(A_D), p(A_D_inv), p(L_D), p(U_D), C.c_int(outer), C.c_int(inner)))

    def init(self, tol):
        r0 = C.c_double()
        self.ctx.check(self.ctx.lib.bis_cg_init(self.ctx.h, self.h, C.c_double(tol), C.byref(r0)))
        return r0.value

    def iterate(self, n):
        self.ctx.check(self.ctx.lib.bis_cg_iterate(self.ctx.h, self.h, C.c_int(n)))

    def status(self, hist_cap=4096):
        iters, conv = C.c_int(), C.c_int()
        hist = np.zeros(hist_cap)
        self.ctx.check(self.ctx.lib.bis_cg_status(self.ctx.h, self.h, C.byref(iters), C.byref(conv),
                                                  hist.ctypes, C.c_int(hist_cap)))
        return iters.value, bool(conv.value), hist[:min(iters.value + 1, hist_cap)].copy()

    def free(self):
        if self.h:
            self.ctx.lib.bis_cg_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()


# ---- multi-GPU (1-D row partition) -------------------------------------------
class Stat:
    """Jacobi / Gauss-Seidel / symmetric Gauss-Seidel as device schedules (bis_stat_*)."""
    KIND = {"j": 0, "gs": 1, "sgs": 2}

    def __init__(self, ctx, kind, A, D, b, x, Ls=None, Us=None):
        self.ctx = ctx
        self.h = C.c_void_p()
        self._keep = (A, D, b, x, Ls, Us)
        ctx.check(ctx.lib.bis_stat_create(ctx.h, C.c_int(self.KIND[kind]), A.h, Ls.h if Ls else None, Us.h if Us else None,
                                          C.c_void_p(D.ptr), C.c_void_p(b.ptr), C.c_void_p(x.ptr), C.byref(self.h)))

    def init(self, tol):
        r0 = C.c_double()
        self.ctx.check(self.ctx.lib.bis_stat_init(self.ctx.h, self.h, C.c_double(tol), C.byref(r0)))
        return r0.value

    def iterate(self, n):
        self.ctx.check(self.ctx.lib.bis_stat_iterate(self.ctx.h, self.h, C.c_int(int(n))))

    def status(self, hist_cap=4096):
        it, conv = C.c_int(), C.c_int()
        hist = np.zeros(hist_cap)
        self.ctx.check(self.ctx.lib.bis_stat_status(self.ctx.h, self.h, C.byref(it), C.byref(conv), hist.ctypes, C.c_int(hist_cap)))
        return it.value, bool(conv.value), hist[:min(it.value + 1, hist_cap)]

    def solution(self, out):
        self.ctx.check(self.ctx.lib.bis_stat_solution(self.ctx.h, self.h, C.c_void_p(out.ptr)))

    def free(self):
        if self.h:
            self.ctx.lib.bis_stat_destroy(self.ctx.h, self.h)
            self.h = None


class CommOps(C.Structure):
    _fields_ = [("user", C.c_void_p),
                ("allreduce_sum", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)),
                ("exchange", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                                         C.c_int))]


def halo_plan(n_local, row_ptr, col_global, n_ranks, rank, row_starts):
    """Host-only planning (bis_halo_plan): returns (halo_cols, recv_counts, interior)."""
    lib = load_library()
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    col = np.ascontiguousarray(col_global, dtype=np.int32)
    rs = np.ascontiguousarray(row_starts, dtype=np.int64)
    n_halo = C.c_int64()
    recv = np.zeros(n_ranks, dtype=np.int64)
    interior = np.zeros(2, dtype=np.int64)
    st = lib.bis_halo_plan(_i64(n_local), rp.ctypes, col.ctypes, C.c_int(n_ranks), C.c_int(rank),
                           rs.ctypes, C.byref(n_halo), None, _i64(0), recv.ctypes, interior.ctypes)
    if st != 0:
        raise BisError(f"bis_halo_plan failed: {st}")
    halo = np.zeros(n_halo.value, dtype=np.int32)
    st = lib.bis_halo_plan(_i64(n_local), rp.ctypes, col.ctypes, C.c_int(n_ranks), C.c_int(rank),
                           rs.ctypes, C.byref(n_halo), halo.ctypes, _i64(halo.size), recv.ctypes,
                           interior.ctypes)
    if st != 0:
        raise BisError(f"bis_halo_plan failed: {st}")
    return halo, recv, interior


class Dist:
    """bis_dist: the row-partitioned operator of one rank."""

    def __init__(self, ctx, A_local, rank, n_ranks, row_starts):
        self.ctx, self.rank, self.n_ranks = ctx, rank, n_ranks
        self.row_starts = np.ascontiguousarray(row_starts, dtype=np.int64)
        self.h = C.c_void_p()
        ctx.check(ctx.lib.bis_dist_create(ctx.h, A_local.h, C.c_int(rank), C.c_int(n_ranks),
                                          self.row_starts.ctypes, C.byref(self.h)))
        A_local.h = C.c_void_p()  # consumed
        nl, ne = C.c_int64(), C.c_int64()
        ctx.lib.bis_dist_vec_len(self.h, C.byref(nl), C.byref(ne))
        self.n_local, self.n_ext = nl.value, ne.value
        self._keep = None

    def halo_info(self):
        n_halo = C.c_int64()
        recv = np.zeros(self.n_ranks, dtype=np.int64)
        self.ctx.lib.bis_dist_halo_info(self.h, C.byref(n_halo), None, _i64(0), recv.ctypes)
        halo = np.zeros(n_halo.value, dtype=np.int32)
        self.ctx.lib.bis_dist_halo_info(self.h, C.byref(n_halo), halo.ctypes, _i64(halo.size),
                                        recv.ctypes)
        return halo, recv

    def set_send_lists(self, send_counts, send_cols):
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        cols = np.ascontiguousarray(send_cols, dtype=np.int32)
        self.ctx.check(self.ctx.lib.bis_dist_set_send_lists(self.ctx.h, self.h, sc.ctypes, cols.ctypes))

    def set_comm(self, ops):
        self._keep = ops  # keep the callbacks alive
        self.ctx.check(self.ctx.lib.bis_dist_set_comm(self.ctx.h, self.h, C.byref(ops)))

    def use_rccl(self, unique_id_bytes):
        buf = C.create_string_buffer(bytes(unique_id_bytes), 128)
        self.ctx.check(self.ctx.lib.bis_dist_use_rccl(self.ctx.h, self.h, buf))

    def spmv(self, x_ext, y):
        self.ctx.check(self.ctx.lib.bis_dist_spmv(self.ctx.h, self.h, C.c_void_p(x_ext.ptr),
                                                  C.c_void_p(y.ptr)))

    def dot(self, a, b):
        out = C.c_double()
        dev = self.ctx.alloc(1)
        self.ctx.check(self.ctx.lib.bis_dist_dot(self.ctx.h, self.h, C.c_void_p(a.ptr),
                                                 C.c_void_p(b.ptr), C.c_void_p(dev.ptr), C.byref(out)))
        dev.free()
        return out.value

    def stats(self):
        nh, ns, ni = C.c_int64(), C.c_int64(), C.c_int64()
        nn, nr = C.c_int(), C.c_int()
        self.ctx.lib.bis_dist_stats(self.h, C.byref(nh), C.byref(ns), C.byref(ni), C.byref(nn), C.byref(nr))
        return dict(halo_entries=nh.value, send_entries=ns.value, halo_bytes_per_spmv=8 * (nh.value + ns.value),
                    interior_rows=ni.value, neighbours=nn.value, rccl_ranks_seen=nr.value)

    def spmv_stream_info(self):
        """(col_bytes, val_bytes, n_dict, form) of the interior rows' SpMV (bis_dist_spmv_stream_info)."""
        c, v, n, f = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(self.ctx.lib.bis_dist_spmv_stream_info(self.ctx.h, self.h, C.byref(c), C.byref(v), C.byref(n), C.byref(f)))
        return c.value, v.value, n.value, f.value

    def spmv_streamed_bytes(self):
        """Bytes this rank's distributed SpMV moves at least per launch triple (bis_dist_spmv_streamed_bytes)."""
        b = C.c_int64()
        self.ctx.check(self.ctx.lib.bis_dist_spmv_streamed_bytes(self.ctx.h, self.h, C.byref(b)))
        return b.value

    def profile_read(self):
        ne, na = C.c_int64(), C.c_int64()
        te, ta = C.c_double(), C.c_double()
        self.ctx.check(self.ctx.lib.bis_dist_profile_read(self.ctx.h, self.h, C.byref(ne), C.byref(te),
                                                          C.byref(na), C.byref(ta)))
        return dict(exchanges=ne.value, exchange_ms=te.value, allreduces=na.value, allreduce_ms=ta.value)

    def cg(self, b, x, A_D=None):
        cg = CG.__new__(CG)
        cg.ctx = self.ctx
        cg.h = C.c_void_p()
        self.ctx.check(self.ctx.lib.bis_dist_cg_create(
            self.ctx.h, self.h, C.c_void_p(A_D.ptr) if A_D else C.c_void_p(), C.c_void_p(b.ptr),
            C.c_void_p(x.ptr), C.byref(cg.h)))
        return cg

    def free(self):
        if self.h:
            self.ctx.lib.bis_dist_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()


def rccl_unique_id(ctx):
    buf = C.create_string_buffer(128)
    ctx.check(ctx.lib.bis_rccl_unique_id(ctx.h, buf))
    return bytes(buf.raw)
