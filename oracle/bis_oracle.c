/*
 * bis_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's SpMV + preconditioner-apply +
 * BLAS-1 hot path and of the solver iteration schedules that call it
 * (DanecLacey/basic_iterative_solvers, mounted at /root/reference).  Every
 * function cites the reference file:line it follows.  Nothing under
 * basic_iterative_solvers_amd/ (the product) may include, link or call this
 * file: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker / reported CPU baseline.
 *
 * Pinning: this restatement is validated (tests/test_oracle_vs_ref.py, run in
 * the build container) against the real reference compiled from its own
 * sources into oracle/_ref/ (oracle/Makefile), and against golden vectors
 * generated from that build and committed under tests/golden/.
 *
 * Arithmetic conventions (stated because the reference leaves them to the
 * compiler): a*b+c patterns are evaluated with fma(), which is what
 * g++ -O3 -march=native emits for the reference on an FMA host and what the
 * HIP compiler emits on gfx950, so elementwise kernels are bit-comparable.
 * Row sums run strictly left to right in CRS storage order (the reference's
 * `omp simd reduction` lets the compiler re-associate; the parity tolerance
 * for that is stated in the tests).  Reductions (dot / norm) follow the
 * reference's per-thread naive partial sums (static schedule) -- with one
 * thread that is a plain left-to-right sum.
 */
#include <ctype.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* PrecondType, reference common.hpp:38-47 (same ordinal values). */
enum {
    ORC_PC_NONE = 0,
    ORC_PC_JACOBI = 1,
    ORC_PC_GS = 2,
    ORC_PC_BGS = 3,
    ORC_PC_SGS = 4,
    ORC_PC_2ST = 5,
    ORC_PC_S2ST = 6,
    ORC_PC_ILU0 = 7
};

/* SolverType, reference common.hpp:49-56. */
enum {
    ORC_S_JACOBI = 0,
    ORC_S_GS = 1,
    ORC_S_SGS = 2,
    ORC_S_GMRES = 3,
    ORC_S_CG = 4,
    ORC_S_BICGSTAB = 5
};

/* CRS view with 64-bit row pointers (reference sparse_matrix.hpp:59-66 uses
 * int32 row_ptr; values are identical whenever they fit, and HPCG-512 needs
 * the wider type -- SURVEY.md section 5, defect 6). */
typedef struct {
    int64_t n_rows, n_cols, nnz;
    const int64_t *row_ptr;
    const int32_t *col;
    const double *val;
} orc_crs;

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* kernels.hpp restatements                                                  */
/* ------------------------------------------------------------------------ */

/* native_spmv, kernels.hpp:22-42: y[r] = sum_k val[k]*x[col[k]]. */
ORC_API void orc_spmv(int64_t n_rows, const int64_t *row_ptr,
                      const int32_t *col, const double *val, const double *x,
                      double *y) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_rows; ++r) {
        double acc = 0.0;
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k)
            acc = fma(val[k], x[col[k]], acc);
        y[r] = acc;
    }
}

/* native_sptrsv, kernels.hpp:54-76: forward solve on a strict-lower CRS plus
 * a separate diagonal; x may alias b (gmres.hpp:173 calls it in place). */
ORC_API void orc_sptrsv(int64_t n_rows, const int64_t *row_ptr,
                        const int32_t *col, const double *val, double *x,
                        const double *D, const double *b) {
    for (int64_t r = 0; r < n_rows; ++r) {
        double acc = 0.0;
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k)
            acc = fma(val[k], x[col[k]], acc);
        x[r] = (b[r] - acc) / D[r];
    }
}

/* native_bsptrsv, kernels.hpp:88-107: backward solve, rows N-1..0. */
ORC_API void orc_bsptrsv(int64_t n_rows, const int64_t *row_ptr,
                         const int32_t *col, const double *val, double *x,
                         const double *D, const double *b) {
    for (int64_t r = n_rows - 1; r >= 0; --r) {
        double acc = 0.0;
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k)
            acc = fma(val[k], x[col[k]], acc);
        x[r] = (b[r] - acc) / D[r];
    }
}

/* subtract_vectors, kernels.hpp:119-126: r = a - s*b. */
ORC_API void orc_subtract_vectors(double *r, const double *a, const double *b,
                                  int64_t n, double s) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) r[i] = fma(-s, b[i], a[i]);
}

/* sum_vectors, kernels.hpp:128-135: r = a + s*b. */
ORC_API void orc_sum_vectors(double *r, const double *a, const double *b,
                             int64_t n, double s) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) r[i] = fma(s, b[i], a[i]);
}

/* elemwise_mult_vectors, kernels.hpp:137-144: r = a*s*b, evaluated (a*s)*b. */
ORC_API void orc_elemwise_mult_vectors(double *r, const double *a,
                                       const double *b, int64_t n, double s) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) r[i] = (a[i] * s) * b[i];
}

/* elemwise_div_vectors, kernels.hpp:146-153: r = a/(s*b). */
ORC_API void orc_elemwise_div_vectors(double *r, const double *a,
                                      const double *b, int64_t n, double s) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) r[i] = a[i] / (s * b[i]);
}

/* compute_residual, kernels.hpp:155-162: tmp = A x ; res = b - tmp. */
ORC_API void orc_compute_residual(const orc_crs *A, const double *x,
                                  const double *b, double *res, double *tmp) {
    orc_spmv(A->n_rows, A->row_ptr, A->col, A->val, x, tmp);
    orc_subtract_vectors(res, b, tmp, A->n_cols, 1.0);
}

/* dot, kernels.hpp:205-212: per-thread naive partial sums over a static
 * schedule, combined in thread order. */
ORC_API double orc_dot(const double *a, const double *b, int64_t n) {
    double sum = 0.0;
#pragma omp parallel for reduction(+ : sum) schedule(static)
    for (int64_t i = 0; i < n; ++i) sum = fma(a[i], b[i], sum);
    return sum;
}

/* euclidean_vec_norm, kernels.hpp:194-203 (empty vector -> 0,
 * tests/test_utilities.cpp:55-59). */
ORC_API double orc_euclidean_vec_norm(const double *v, int64_t n) {
    double sum = 0.0;
#pragma omp parallel for reduction(+ : sum) schedule(static)
    for (int64_t i = 0; i < n; ++i) sum = fma(v[i], v[i], sum);
    return sqrt(sum);
}

/* scale, kernels.hpp:214-220. */
ORC_API void orc_scale(double *r, const double *v, double s, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) r[i] = v[i] * s;
}

/* init_vector, kernels.hpp:236-241. */
ORC_API void orc_init_vector(double *v, double val, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) v[i] = val;
}

/* copy_vector, kernels.hpp:252-257. */
ORC_API void orc_copy_vector(double *out, const double *in, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = in[i];
}

/* dgemm_transpose1 as called at gmres.hpp:358 (kernels.hpp:259-271 with
 * n_cols_B == 1): out[i] = sum_{k<n_vec} V[k*N+i]*y[k].  The reference passes
 * n_vec = n_solver_iters+1 and at a restart reads y[m] one past the end of y
 * (SURVEY.md section 5 defect 1); the defined semantics used here and in the
 * product is y[n_solver_iters] == 0, i.e. the caller passes n_vec =
 * n_solver_iters. */
ORC_API void orc_multi_axpy(const double *V, int64_t N, const double *y,
                            int n_vec, double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        double acc = 0.0;
        for (int k = 0; k < n_vec; ++k) acc = fma(V[(int64_t)k * N + i], y[k], acc);
        out[i] = acc;
    }
}

/* normalize_x, methods/jacobi.hpp:27-40: x_new = (b - (x_new - D*x_old))/D. */
ORC_API void orc_normalize_x(double *x_new, const double *x_old,
                             const double *D, const double *b, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double adjusted = fma(-D[i], x_old[i], x_new[i]);
        x_new[i] = (b[i] - adjusted) / D[i];
    }
}

/* two_stage_gauss_seidel, kernels.hpp:312-333.  `inner_iters` is the
 * reference's compile-time PRECOND_INNER_ITERS.  tmp/work are swapped by
 * value exactly as the reference does (the caller's pointers are unchanged).
 * input may alias output (kernels.hpp:384). */
ORC_API void orc_two_stage_gs(const orc_crs *strict, double *tmp, double *work,
                              const double *D_inv, const double *input,
                              double *output, int64_t n, int inner_iters) {
    orc_elemwise_mult_vectors(work, D_inv, input, n, 1.0);
    orc_copy_vector(output, work, n);
    for (int inner = 1; inner <= inner_iters; ++inner) {
        orc_spmv(strict->n_rows, strict->row_ptr, strict->col, strict->val,
                 work, tmp);
        orc_elemwise_mult_vectors(tmp, D_inv, tmp, n, -1.0);
        double *t = work;
        work = tmp;
        tmp = t;
        orc_sum_vectors(output, output, work, n, 1.0);
    }
}

/* apply_preconditioner, kernels.hpp:336-414: z = M^{-1} y. */
ORC_API void orc_apply_preconditioner(int pc, int64_t n, const orc_crs *Ls,
                                      const orc_crs *Us, const double *A_D,
                                      const double *A_D_inv, const double *L_D,
                                      const double *U_D, double *out,
                                      double *in, double *tmp, double *work,
                                      int outer_iters, int inner_iters) {
    double *saved = NULL;
    if (outer_iters > 1) {
        saved = (double *)malloc(sizeof(double) * (size_t)n);
        orc_copy_vector(saved, in, n);
    }
    for (int it = 0; it < outer_iters; ++it) {
        switch (pc) {
        case ORC_PC_JACOBI:
            orc_elemwise_div_vectors(out, in, A_D, n, 1.0);
            break;
        case ORC_PC_GS:
            orc_sptrsv(n, Ls->row_ptr, Ls->col, Ls->val, out, A_D, in);
            break;
        case ORC_PC_BGS:
            orc_bsptrsv(n, Us->row_ptr, Us->col, Us->val, out, A_D, in);
            break;
        case ORC_PC_SGS:
            orc_sptrsv(n, Ls->row_ptr, Ls->col, Ls->val, tmp, A_D, in);
            orc_elemwise_mult_vectors(tmp, tmp, A_D, n, 1.0);
            orc_bsptrsv(n, Us->row_ptr, Us->col, Us->val, out, A_D, tmp);
            break;
        case ORC_PC_2ST:
            orc_two_stage_gs(Ls, tmp, work, A_D_inv, in, out, n, inner_iters);
            break;
        case ORC_PC_S2ST:
            orc_two_stage_gs(Ls, tmp, work, A_D_inv, in, out, n, inner_iters);
            orc_elemwise_mult_vectors(out, out, A_D, n, 1.0);
            orc_two_stage_gs(Us, tmp, work, A_D_inv, out, out, n, inner_iters);
            break;
        case ORC_PC_ILU0:
            orc_sptrsv(n, Ls->row_ptr, Ls->col, Ls->val, tmp, L_D, in);
            orc_bsptrsv(n, Us->row_ptr, Us->col, Us->val, out, U_D, tmp);
            break;
        default:
            orc_copy_vector(out, in, n);
        }
        if (outer_iters > 1 && it != outer_iters - 1) orc_copy_vector(in, out, n);
    }
    if (outer_iters > 1) orc_copy_vector(in, saved, n);
    free(saved);
}

/* ------------------------------------------------------------------------ */
/* Setup restatements (host-side, run once)                                  */
/* ------------------------------------------------------------------------ */

/* MatrixCOO::read_from_mtx (default reader), sparse_matrix.hpp:261-350 on top
 * of mm_read_unsymmetric_sparse (utilities/mmio.hpp:123-210): 1-based ->
 * 0-based, pattern entries 0.01, symmetric files expanded with the mirrored
 * entry inserted directly after its source entry, then a STABLE sort by row
 * only (sort_perm, sparse_matrix.hpp:20-30), so the column order within a row
 * is file order.  Followed by convert_coo_to_crs (utilities.hpp:326-367).
 * Returns 0 on success; arrays are malloc'ed and owned by the caller
 * (orc_free). */
ORC_API int orc_read_mtx_crs(const char *path, int64_t *n_rows, int64_t *n_cols,
                             int64_t *nnz_out, int64_t **row_ptr_out,
                             int32_t **col_out, double **val_out) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char line[1024];
    if (!fgets(line, sizeof line, f)) { fclose(f); return -2; }
    char banner[64], object[64], format[64], field[64], symmetry[64];
    if (sscanf(line, "%63s %63s %63s %63s %63s", banner, object, format, field,
               symmetry) != 5 || strcmp(banner, "%%MatrixMarket") != 0) {
        fclose(f);
        return -2;
    }
    for (char *p = object; *p; ++p) *p = (char)tolower(*p);
    for (char *p = format; *p; ++p) *p = (char)tolower(*p);
    for (char *p = field; *p; ++p) *p = (char)tolower(*p);
    for (char *p = symmetry; *p; ++p) *p = (char)tolower(*p);
    int is_pattern = strcmp(field, "pattern") == 0;
    int is_real = strcmp(field, "real") == 0 || strcmp(field, "integer") == 0;
    int is_sym = strcmp(symmetry, "symmetric") == 0;
    int is_gen = strcmp(symmetry, "general") == 0;
    if (strcmp(object, "matrix") != 0 || strcmp(format, "coordinate") != 0 ||
        !(is_pattern || is_real) || !(is_sym || is_gen)) {
        fclose(f);
        return -3; /* unsupported, sparse_matrix.hpp:281-287 */
    }
    long M = 0, N = 0, nz = 0;
    for (;;) {
        if (!fgets(line, sizeof line, f)) { fclose(f); return -4; }
        if (line[0] == '%') continue;
        if (sscanf(line, "%ld %ld %ld", &M, &N, &nz) == 3) break;
    }
    if (M != N) { fclose(f); return -5; } /* sparse_matrix.hpp:297-299 */

    int64_t cap = is_sym ? 2 * (int64_t)nz : (int64_t)nz;
    int32_t *I = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cap ? cap : 1));
    int32_t *J = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cap ? cap : 1));
    double *V = (double *)malloc(sizeof(double) * (size_t)(cap ? cap : 1));
    int64_t cnt = 0;
    for (long e = 0; e < nz; ++e) {
        int i, j;
        double v = 0.01;
        if (is_pattern) {
            if (fscanf(f, "%d %d", &i, &j) != 2) { fclose(f); return -6; }
        } else {
            if (fscanf(f, "%d %d %lg", &i, &j, &v) != 3) { fclose(f); return -6; }
        }
        --i;
        --j;
        I[cnt] = i; J[cnt] = j; V[cnt] = v; ++cnt;
        if (is_sym && i != j) { I[cnt] = j; J[cnt] = i; V[cnt] = v; ++cnt; }
    }
    fclose(f);

    /* stable counting sort by row == std::stable_sort on the row key */
    int64_t *rp = (int64_t *)calloc((size_t)M + 1, sizeof(int64_t));
    for (int64_t e = 0; e < cnt; ++e) rp[I[e] + 1]++;
    for (long r = 0; r < M; ++r) rp[r + 1] += rp[r];
    int32_t *col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cnt ? cnt : 1));
    double *val = (double *)malloc(sizeof(double) * (size_t)(cnt ? cnt : 1));
    int64_t *cursor = (int64_t *)malloc(sizeof(int64_t) * (size_t)(M ? M : 1));
    for (long r = 0; r < M; ++r) cursor[r] = rp[r];
    for (int64_t e = 0; e < cnt; ++e) {
        int64_t p = cursor[I[e]]++;
        col[p] = J[e];
        val[p] = V[e];
    }
    free(cursor); free(I); free(J); free(V);
    *n_rows = M; *n_cols = N; *nnz_out = cnt;
    *row_ptr_out = rp; *col_out = col; *val_out = val;
    return 0;
}

ORC_API void orc_free(void *p) { free(p); }

/* convert_coo_to_crs, utilities.hpp:326-367: requires row-sorted COO; col/val
 * are copied through unchanged, row_ptr is the prefix sum of per-row counts. */
ORC_API int orc_coo_to_crs(int64_t n_rows, int64_t nnz, const int32_t *I,
                           const int32_t *J, const double *V, int64_t *row_ptr,
                           int32_t *col, double *val) {
    for (int64_t e = 0; e < nnz; ++e) { col[e] = J[e]; val[e] = V[e]; }
    for (int64_t r = 0; r <= n_rows; ++r) row_ptr[r] = 0;
    for (int64_t e = 0; e < nnz; ++e) row_ptr[I[e] + 1]++;
    for (int64_t r = 0; r < n_rows; ++r) row_ptr[r + 1] += row_ptr[r];
    return row_ptr[n_rows] == nnz ? 0 : 1;
}

/* split_LU (split_LU_new), utilities/LU_factors.hpp:122-309: A -> L (col<=i),
 * L_strict (col<i), U (col>=i), U_strict (col>i), row order preserved.
 * Two-call protocol: counts first (arrays NULL), then fill. */
ORC_API void orc_split_LU_count(const orc_crs *A, int64_t *nnz4) {
    int64_t l = 0, ls = 0, u = 0, us = 0;
    for (int64_t i = 0; i < A->n_rows; ++i)
        for (int64_t k = A->row_ptr[i]; k < A->row_ptr[i + 1]; ++k) {
            int32_t c = A->col[k];
            if (c < i) { ++l; ++ls; }
            if (c == i) { ++l; ++u; }
            if (c > i) { ++u; ++us; }
        }
    nnz4[0] = l; nnz4[1] = ls; nnz4[2] = u; nnz4[3] = us;
}

ORC_API void orc_split_LU_fill(const orc_crs *A, int64_t *L_rp, int32_t *L_col,
                               double *L_val, int64_t *Ls_rp, int32_t *Ls_col,
                               double *Ls_val, int64_t *U_rp, int32_t *U_col,
                               double *U_val, int64_t *Us_rp, int32_t *Us_col,
                               double *Us_val) {
    int64_t l = 0, ls = 0, u = 0, us = 0;
    L_rp[0] = Ls_rp[0] = U_rp[0] = Us_rp[0] = 0;
    for (int64_t i = 0; i < A->n_rows; ++i) {
        for (int64_t k = A->row_ptr[i]; k < A->row_ptr[i + 1]; ++k) {
            int32_t c = A->col[k];
            double v = A->val[k];
            if (c < i) {
                L_col[l] = c; L_val[l++] = v;
                Ls_col[ls] = c; Ls_val[ls++] = v;
            }
            if (c == i) {
                L_col[l] = c; L_val[l++] = v;
                U_col[u] = c; U_val[u++] = v;
            }
            if (c > i) {
                U_col[u] = c; U_val[u++] = v;
                Us_col[us] = c; Us_val[us++] = v;
            }
        }
        L_rp[i + 1] = l; Ls_rp[i + 1] = ls; U_rp[i + 1] = u; Us_rp[i + 1] = us;
    }
}

/* peel_diag_crs (peel_diag_crs_new), utilities/LU_factors.hpp:827-869:
 * extract D (and 1/D), swap the diagonal entry to the row's last slot.
 * Returns 0, or 1+row for a zero diagonal (|d|<1e-16, common.hpp:388-391),
 * or -(1+row) for a missing one (common.hpp:393-396); the reference exits. */
ORC_API int64_t orc_peel_diag_crs(int64_t n_rows, const int64_t *row_ptr,
                                  int32_t *col, double *val, double *D,
                                  double *D_inv) {
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t start = row_ptr[r], last = row_ptr[r + 1] - 1, dj = -1;
        for (int64_t j = start; j <= last; ++j)
            if (col[j] == r) {
                dj = j;
                D[r] = val[j];
                if (fabs(D[r]) < 1e-16) return 1 + r;
                if (D_inv) D_inv[r] = 1.0 / D[r];
            }
        if (dj < 0) return -(1 + r);
        if (dj != last) {
            int32_t tc = col[dj]; col[dj] = col[last]; col[last] = tc;
            double tv = val[dj]; val[dj] = val[last]; val[last] = tv;
        }
    }
    return 0;
}

/* extract_scale, utilities/LU_factors.hpp:880-898: s = 1/sqrt(|a_ii|). */
ORC_API int64_t orc_extract_scale(const orc_crs *A, double *D_scale) {
    for (int64_t r = 0; r < A->n_rows; ++r)
        for (int64_t j = A->row_ptr[r]; j < A->row_ptr[r + 1]; ++j)
            if (A->col[j] == r) {
                if (fabs(A->val[j]) < 1e-16) return 1 + r;
                D_scale[r] = 1.0 / sqrt(fabs(A->val[j]));
            }
    return 0;
}

/* scale_mat / scale_vec, preprocessing.hpp:9-24: a_ij *= (s_i*s_j). */
ORC_API void orc_scale_mat(int64_t n_rows, const int64_t *row_ptr,
                           const int32_t *col, double *val, const double *s) {
    for (int64_t r = 0; r < n_rows; ++r) {
        double sr = s[r];
        for (int64_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j)
            val[j] *= (sr * s[col[j]]);
    }
}
ORC_API void orc_scale_vec(double *v, const double *s, int64_t n) {
    for (int64_t i = 0; i < n; ++i) v[i] = s[i] * v[i];
}

/* factor_ILU0_old, utilities/LU_factors.hpp:320-539 (the serial algorithm;
 * the wired-in factor_ILU0_new needs SMAX -- SURVEY.md section 5 defect 2).
 * Row-by-row IKJ elimination restricted to A's pattern:
 *   - dependencies k<i processed in ascending column order (:348, :355-357)
 *   - pivot |u_kk| < 1e-16 skips the elimination step (:369-370)
 *   - an update hits only positions whose workspace value is non-zero (:383)
 *   - |u_ii| < pivot_tol is replaced by sign(u_ii)*pivot_repl (:410-412)
 * Outputs: L_strict (ascending columns), L_D == 1, U_strict (ascending
 * columns, via split_LU :538) and U_D.  Arrays sized like A's strict parts
 * (counts from orc_split_LU_count).  Returns 0. */
typedef struct { int32_t c; double v; } orc_cv;
static int orc_cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}
ORC_API int orc_factor_ilu0(const orc_crs *A, double pivot_tol,
                            double pivot_repl, int64_t *Ls_rp, int32_t *Ls_col,
                            double *Ls_val, double *L_D, int64_t *Us_rp,
                            int32_t *Us_col, double *Us_val, double *U_D) {
    int64_t n = A->n_rows;
    double *w = (double *)calloc((size_t)(n ? n : 1), sizeof(double));
    int64_t maxrow = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t len = A->row_ptr[i + 1] - A->row_ptr[i];
        if (len > maxrow) maxrow = len;
    }
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(maxrow + 1));
    int64_t lpos = 0, upos = 0;
    Ls_rp[0] = 0;
    Us_rp[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t len = 0;
        for (int64_t p = A->row_ptr[i]; p < A->row_ptr[i + 1]; ++p) {
            w[A->col[p]] = A->val[p];
            idx[len++] = A->col[p];
        }
        qsort(idx, (size_t)len, sizeof(int32_t), orc_cmp_i32);
        for (int64_t q = 0; q < len; ++q) {
            int32_t k = idx[q];
            if (k >= i) break;
            double pivot = U_D[k]; /* the (k,k) entry of the finished row k */
            if (fabs(pivot) < 1e-16) continue;
            double factor = w[k] / pivot;
            w[k] = factor;
            /* finished U row k = strict part (ascending) then the diagonal */
            for (int64_t p = Us_rp[k]; p < Us_rp[k + 1]; ++p) {
                int32_t j = Us_col[p];
                if (w[j] != 0.0) w[j] -= factor * Us_val[p];
            }
        }
        double u_diag = 0.0;
        for (int64_t q = 0; q < len; ++q) {
            int32_t j = idx[q];
            if (j < i) { Ls_col[lpos] = j; Ls_val[lpos++] = w[j]; }
            else if (j == i) u_diag = w[j];
            else { Us_col[upos] = j; Us_val[upos++] = w[j]; }
        }
        if (fabs(u_diag) < pivot_tol)
            u_diag = (u_diag >= 0 ? 1.0 : -1.0) * pivot_repl;
        U_D[i] = u_diag;
        L_D[i] = 1.0;
        Ls_rp[i + 1] = lpos;
        Us_rp[i + 1] = upos;
        for (int64_t q = 0; q < len; ++q) w[idx[q]] = 0.0;
    }
    free(idx);
    free(w);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Synthetic inputs -- definitions shared, as a        */
/* specification, with the product's device generators.                      */
/* ------------------------------------------------------------------------ */

static inline int64_t hpcg_c(int64_t j, int64_t n) { /* stencil extent at j */
    return 1 + (j > 0) + (j < n - 1);
}
static inline int64_t hpcg_f(int64_t k, int64_t n) { /* sum_{j<k} c(j,n) */
    int64_t a = k - 1 > 0 ? k - 1 : 0;
    int64_t b = k < n - 1 ? k : n - 1;
    return k + a + b;
}

/* nnz of rows [0,row) of the HPCG matrix -- closed form, 64-bit. */
ORC_API int64_t orc_hpcg_row_ptr(int64_t row, int64_t nx, int64_t ny,
                                 int64_t nz) {
    int64_t total_rows = nx * ny * nz;
    int64_t Sx = hpcg_f(nx, nx), Sy = hpcg_f(ny, ny);
    if (row >= total_rows) return hpcg_f(nz, nz) * Sy * Sx;
    int64_t x = row % nx, y = (row / nx) % ny, z = row / (nx * ny);
    return hpcg_f(z, nz) * Sy * Sx +
           hpcg_c(z, nz) * (hpcg_f(y, ny) * Sx + hpcg_c(y, ny) * hpcg_f(x, nx));
}

/* HPCG 27-point operator on an nx*ny*nz grid (x fastest): a_ii = 26,
 * a_ij = -1 for in-grid neighbours, open boundaries, ascending columns.
 * Generates global rows [row0,row1) with GLOBAL column indices;
 * row_ptr[0] = 0 (local offsets). */
ORC_API void orc_gen_hpcg(int64_t nx, int64_t ny, int64_t nz, int64_t row0,
                          int64_t row1, int64_t *row_ptr, int32_t *col,
                          double *val) {
    int64_t base = orc_hpcg_row_ptr(row0, nx, ny, nz);
#pragma omp parallel for schedule(static)
    for (int64_t row = row0; row < row1; ++row) {
        int64_t x = row % nx, y = (row / nx) % ny, z = row / (nx * ny);
        int64_t p = orc_hpcg_row_ptr(row, nx, ny, nz) - base;
        row_ptr[row - row0] = p;
        for (int dz = -1; dz <= 1; ++dz) {
            if (z + dz < 0 || z + dz >= nz) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                if (y + dy < 0 || y + dy >= ny) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    if (x + dx < 0 || x + dx >= nx) continue;
                    col[p] = (int32_t)(row + dx + nx * (dy + ny * (int64_t)dz));
                    val[p] = (dx == 0 && dy == 0 && dz == 0) ? 26.0 : -1.0;
                    ++p;
                }
            }
        }
    }
    row_ptr[row1 - row0] = orc_hpcg_row_ptr(row1, nx, ny, nz) - base;
}

/* Counter-based uniform [0,1): splitmix64 finaliser of (seed, i). */
static inline double anderson_u01(uint64_t seed, uint64_t i) {
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + (i + 1) * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
ORC_API double orc_anderson_diag(uint64_t seed, uint64_t i, double W,
                                 double shift) {
    return fma(W, anderson_u01(seed, i) - 0.5, shift);
}

/* FEM-like unstructured stand-in for SuiteSparse Flan_1565 (SURVEY.md section 8d-3;
 * the .mtx itself is not fetchable): nx*ny*nz nodes, 3 unknowns per node, row =
 * 3*node + d.  Node a couples to itself and to each in-grid neighbour b of its
 * 27-point neighbourhood that survives a symmetric coin flip
 * u01(seed^K1, min*n_nodes+max)*100 < keep; a coupling is a full 3x3 block.
 * Off-diagonal a_rc = -(0.05 + 0.95*u01(seed^K2, min(r,c)*n_rows + max(r,c)))
 * (symmetric); a_rr = 1 + sum_c |a_rc| accumulated left to right in column
 * order (strictly diagonally dominant => SPD).  Ascending columns; about
 * 3*(1 + 26*keep/100) non-zeros per interior row (keep = 85: ~69). */
#define FEM_K1 0x5851F42D4C957F2Dull
#define FEM_K2 0x14057B7EF767814Full
static inline int fem_kept(uint64_t seed, int64_t a, int64_t b, int64_t n_nodes, int keep) {
    const int64_t lo = a < b ? a : b, hi = a < b ? b : a;
    return anderson_u01(seed ^ FEM_K1, (uint64_t)(lo * n_nodes + hi)) * 100.0 < (double)keep;
}
static inline double fem_offdiag(uint64_t seed, int64_t r, int64_t c, int64_t n_rows) {
    const int64_t lo = r < c ? r : c, hi = r < c ? c : r;
    return -(0.05 + 0.95 * anderson_u01(seed ^ FEM_K2, (uint64_t)(lo * n_rows + hi)));
}
/* one row: returns its length; writes columns/values when col != NULL */
static int64_t fem_row(int64_t nx, int64_t ny, int64_t nz, int keep, uint64_t seed, int64_t row,
                       int32_t *col, double *val) {
    const int64_t n_nodes = nx * ny * nz, n_rows = 3 * n_nodes;
    const int64_t a = row / 3;
    const int64_t i = a % nx, j = (a / nx) % ny, k = a / (nx * ny);
    int64_t len = 0, dpos = -1;
    double acc = 0.0;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int64_t ii = i + dx, jj = j + dy, kk = k + dz;
                if (ii < 0 || ii >= nx || jj < 0 || jj >= ny || kk < 0 || kk >= nz) continue;
                const int64_t b = ii + nx * (jj + ny * kk);
                if (b != a && !fem_kept(seed, a, b, n_nodes, keep)) continue;
                for (int d = 0; d < 3; ++d) {
                    const int64_t c = 3 * b + d;
                    if (col) {
                        col[len] = (int32_t)c;
                        if (c == row) { dpos = len; val[len] = 0.0; }
                        else { const double v = fem_offdiag(seed, row, c, n_rows); val[len] = v; acc += fabs(v); }
                    }
                    ++len;
                }
            }
    if (col) val[dpos] = 1.0 + acc;
    return len;
}
/* rows [row0,row1): pass 1 (col == NULL) fills row_ptr[0..n] (local offsets) and
 * returns nnz; pass 2 fills col/val at those offsets. */
ORC_API int64_t orc_gen_fem(int64_t nx, int64_t ny, int64_t nz, int keep, uint64_t seed, int64_t row0,
                            int64_t row1, int64_t *row_ptr, int32_t *col, double *val) {
    const int64_t n = row1 - row0;
    if (!col) {
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < n; ++r) row_ptr[r + 1] = fem_row(nx, ny, nz, keep, seed, row0 + r, NULL, NULL);
        row_ptr[0] = 0;
        for (int64_t r = 0; r < n; ++r) row_ptr[r + 1] += row_ptr[r];
        return row_ptr[n];
    }
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) fem_row(nx, ny, nz, keep, seed, row0 + r, col + row_ptr[r], val + row_ptr[r]);
    return row_ptr[n];
}

/* Unstructured input for config 5 (stands in for reading SuiteSparse Flan_1565.mtx through
 * sparse_matrix.hpp:225-357; the file is not fetchable): the FEM-like matrix above under a
 * seeded random symmetric permutation of its ROWS (not nodes), B = P A P^T, columns ascending
 * inside a row -- no grid, no node blocks, no locality: what the reference's reader hands the
 * solvers for a mesh whose numbering carries no structure.  perm[new] = old is the stable
 * ascending order of the 64-bit keys hash(seed ^ K3, old) (ties, which a 64-bit hash makes
 * practically impossible, keep the lower old index first).  The values are those of orc_gen_fem
 * (the diagonal keeps the sum accumulated in A's column order). */
#define FEM_K3 0x2545F4914F6CDD1Dull
static inline uint64_t unstr_key(uint64_t seed, uint64_t i) {
    uint64_t z = (seed ^ FEM_K3) * 0x9E3779B97F4A7C15ull + (i + 1) * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
typedef struct { uint64_t key; int64_t idx; } unstr_pair;
static int unstr_cmp(const void *a, const void *b) {
    const unstr_pair *x = (const unstr_pair *)a, *y = (const unstr_pair *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : x->idx > y->idx;
}
ORC_API int orc_unstr_perm(int64_t n, uint64_t seed, int32_t *perm) {
    unstr_pair *p = (unstr_pair *)malloc(sizeof(unstr_pair) * (size_t)(n > 0 ? n : 1));
    if (!p) return 1;
    for (int64_t i = 0; i < n; ++i) { p[i].key = unstr_key(seed, (uint64_t)i); p[i].idx = i; }
    qsort(p, (size_t)n, sizeof(unstr_pair), unstr_cmp);
    for (int64_t i = 0; i < n; ++i) perm[i] = (int32_t)p[i].idx;
    free(p);
    return 0;
}
typedef struct { int32_t c; double v; } unstr_ent;
static int unstr_ent_cmp(const void *a, const void *b) {
    const int32_t x = ((const unstr_ent *)a)->c, y = ((const unstr_ent *)b)->c;
    return x < y ? -1 : x > y;
}
/* pass 1 (col == NULL): row_ptr[0..n], returns nnz; pass 2: col / val at those offsets.  perm = orc_unstr_perm. */
ORC_API int64_t orc_gen_unstr(int64_t nx, int64_t ny, int64_t nz, int keep, uint64_t seed, const int32_t *perm,
                              int64_t *row_ptr, int32_t *col, double *val) {
    const int64_t n = 3 * nx * ny * nz;
    if (!col) {
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < n; ++r) row_ptr[r + 1] = fem_row(nx, ny, nz, keep, seed, perm[r], NULL, NULL);
        row_ptr[0] = 0;
        for (int64_t r = 0; r < n; ++r) row_ptr[r + 1] += row_ptr[r];
        return row_ptr[n];
    }
    int32_t *inv = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    if (!inv) return -1;
    for (int64_t r = 0; r < n; ++r) inv[perm[r]] = (int32_t)r;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        int32_t c[81];
        double v[81];
        unstr_ent e[81];
        const int64_t len = fem_row(nx, ny, nz, keep, seed, perm[r], c, v);
        for (int64_t k = 0; k < len; ++k) { e[k].c = inv[c[k]]; e[k].v = v[k]; }
        qsort(e, (size_t)len, sizeof(unstr_ent), unstr_ent_cmp); /* distinct columns: no ties */
        for (int64_t k = 0; k < len; ++k) { col[row_ptr[r] + k] = e[k].c; val[row_ptr[r] + k] = e[k].v; }
    }
    free(inv);
    return row_ptr[n];
}

/* Anderson-L: 7-point, periodic L^3 grid (L >= 3), off-diagonals -t,
 * diagonal W*(u-1/2)+shift with u = u01(seed,row); ascending columns; exactly
 * 7 nnz per row.  Generates rows [row0,row1), global columns. */
ORC_API void orc_gen_anderson(int64_t L, double t, double W, double shift,
                              uint64_t seed, int64_t row0, int64_t row1,
                              int64_t *row_ptr, int32_t *col, double *val) {
#pragma omp parallel for schedule(static)
    for (int64_t row = row0; row < row1; ++row) {
        int64_t x = row % L, y = (row / L) % L, z = row / (L * L);
        int64_t c[7];
        double v[7];
        int64_t xm = (x + L - 1) % L, xp = (x + 1) % L;
        int64_t ym = (y + L - 1) % L, yp = (y + 1) % L;
        int64_t zm = (z + L - 1) % L, zp = (z + 1) % L;
        c[0] = x + L * (y + L * zm);
        c[1] = x + L * (ym + L * z);
        c[2] = xm + L * (y + L * z);
        c[3] = row;
        c[4] = xp + L * (y + L * z);
        c[5] = x + L * (yp + L * z);
        c[6] = x + L * (y + L * zp);
        for (int k = 0; k < 7; ++k) v[k] = -t;
        v[3] = orc_anderson_diag(seed, (uint64_t)row, W, shift);
        for (int a = 1; a < 7; ++a) { /* insertion sort by column */
            int64_t ck = c[a];
            double vk = v[a];
            int b = a - 1;
            while (b >= 0 && c[b] > ck) { c[b + 1] = c[b]; v[b + 1] = v[b]; --b; }
            c[b + 1] = ck;
            v[b + 1] = vk;
        }
        int64_t p = (row - row0) * 7;
        row_ptr[row - row0] = p;
        for (int k = 0; k < 7; ++k) { col[p + k] = (int32_t)c[k]; val[p + k] = v[k]; }
    }
    row_ptr[row1 - row0] = (row1 - row0) * 7;
}

/* ------------------------------------------------------------------------ */
/* Solver iteration schedules                                                */
/* ------------------------------------------------------------------------ */

typedef struct {
    int solver;           /* ORC_S_*   (utilities.hpp:28-52) */
    int precond;          /* ORC_PC_*  (utilities.hpp:69-98) */
    int max_iters;        /* MAX_ITERS      CMakeLists.txt:19 */
    double tol;           /* TOL            CMakeLists.txt:20 */
    int restart_len;      /* Args::restart_length common.hpp:109 */
    int outer_iters;      /* PRECOND_OUTER_ITERS */
    int inner_iters;      /* PRECOND_INNER_ITERS */
    double init_x;        /* INIT_X_VAL */
    double b_val;         /* B_VAL */
    int num_scale;        /* -scale, preprocessing.hpp:39-50 */
    double ilu_pivot_tol; /* ILU0_PIVOT_TOLERANCE */
    double ilu_pivot_repl;
    int ilu_real;         /* 0: the literal reference (ILU0 silently falls back
                             to strict parts of A, defect 2); 1: real serial
                             ILU(0) factors (factor_ILU0_old) */
} orc_opts;

typedef struct {
    int iters;      /* iter_count as printed (GMRES: + restart count) */
    int n_hist;     /* collected_residual_norms_count */
    int converged;
    double stopping_criteria;
    double final_true_residual; /* history[count+1], solver.hpp:158 */
} orc_result;

typedef struct {
    orc_crs A, Ls, Us;
    int64_t N;
    double *x_star, *x_0, *b, *tmp, *work, *residual, *residual_0;
    double *A_D, *A_D_inv, *A_D_scale, *L_D, *U_D;
    double *hist;
    int hist_count, iter_count, restart_count;
    double residual_norm, stopping;
    const orc_opts *o;
} orc_state;

static double *dalloc(int64_t n) {
    return (double *)calloc((size_t)(n ? n : 1), sizeof(double));
}

static void st_precond(orc_state *s, double *out, double *in) {
    orc_apply_preconditioner(s->o->precond, s->N, &s->Ls, &s->Us, s->A_D,
                             s->A_D_inv, s->L_D, s->U_D, out, in, s->tmp,
                             s->work, s->o->outer_iters, s->o->inner_iters);
}
static void st_spmv(orc_state *s, const double *x, double *y) {
    orc_spmv(s->A.n_rows, s->A.row_ptr, s->A.col, s->A.val, x, y);
}
static void swapd(double **a, double **b) { double *t = *a; *a = *b; *b = t; }

/* check_stopping_criteria, solver.hpp:177-192 */
static int st_stop(const orc_state *s) {
    int conv = fabs(s->residual_norm) < s->stopping;
    int over = s->iter_count >= (s->o->max_iters - s->restart_count);
    int div = fabs(s->residual_norm) > DBL_MAX || isnan(s->residual_norm);
    return conv || over || div;
}

/* Solver::save_x_star, solver.hpp:153-159 (after the subclass swap). */
static double st_final_residual(orc_state *s) {
    orc_compute_residual(&s->A, s->x_star, s->b, s->residual, s->tmp);
    return orc_euclidean_vec_norm(s->residual, s->N);
}

/* ---- CG: methods/cg.hpp:6-54 (iteration), :100-118 (init_residual),
 * :129-133 (exchange), :157-166 (save_x_star, record_residual_norm) ---- */
static void run_cg(orc_state *s, orc_result *res) {
    int64_t N = s->N;
    double *x_new = dalloc(N), *x_old = dalloc(N), *p_new = dalloc(N),
           *p_old = dalloc(N), *r_new = dalloc(N), *r_old = dalloc(N),
           *z_new = dalloc(N), *z_old = dalloc(N);
    orc_copy_vector(x_old, s->x_0, N);
    orc_compute_residual(&s->A, x_old, s->b, s->residual, s->tmp);
    st_precond(s, z_old, s->residual);
    orc_copy_vector(p_old, z_old, N);
    orc_copy_vector(r_old, s->residual, N);
    s->residual_norm = orc_euclidean_vec_norm(s->residual, N);
    orc_copy_vector(s->residual_0, s->residual, N);
    s->hist[s->hist_count++] = s->residual_norm;
    s->stopping = s->o->tol * s->residual_norm;
    do {
        st_spmv(s, p_old, s->tmp);
        double tmp_dot = orc_dot(r_old, z_old, N);
        double alpha = tmp_dot / orc_dot(s->tmp, p_old, N);
        orc_sum_vectors(x_new, x_old, p_old, N, alpha);
        orc_subtract_vectors(r_new, r_old, s->tmp, N, alpha);
        st_precond(s, z_new, r_new);
        double beta = orc_dot(r_new, z_new, N) / tmp_dot;
        orc_sum_vectors(p_new, z_new, p_old, N, beta);
        ++s->iter_count;
        s->residual_norm = orc_euclidean_vec_norm(r_new, N);
        s->hist[s->hist_count++] = s->residual_norm;
        swapd(&p_old, &p_new);
        swapd(&z_old, &z_new);
        swapd(&r_old, &r_new);
        swapd(&x_old, &x_new);
    } while (!st_stop(s));
    res->converged = s->residual_norm < s->stopping;
    orc_copy_vector(s->x_star, x_old, N);
    res->final_true_residual = st_final_residual(s);
    free(x_new); free(x_old); free(p_new); free(p_old);
    free(r_new); free(r_old); free(z_new); free(z_old);
}

/* ---- BiCGSTAB: methods/bicgstab.hpp:8-83, :147-169 (init_residual: note
 * residual_0 and p_0 are the PRECONDITIONED initial residual and
 * rho_0 = (r_0, M^{-1} r_0)), :171-185, :215-223 ---- */
static void run_bicgstab(orc_state *s, orc_result *res) {
    int64_t N = s->N;
    double *x_new = dalloc(N), *x_old = dalloc(N), *p_new = dalloc(N),
           *p_old = dalloc(N), *r_new = dalloc(N), *r_old = dalloc(N),
           *v = dalloc(N), *h = dalloc(N), *sv = dalloc(N), *s_tmp = dalloc(N),
           *y = dalloc(N), *z = dalloc(N);
    double *residual = s->residual;
    orc_copy_vector(x_old, s->x_0, N);
    orc_compute_residual(&s->A, x_old, s->b, residual, s->tmp);
    orc_copy_vector(r_old, residual, N);
    s->residual_norm = orc_euclidean_vec_norm(residual, N);
    st_precond(s, residual, residual);
    orc_copy_vector(p_old, residual, N);
    double rho_old = orc_dot(r_old, residual, N), rho_new = 0.0;
    orc_copy_vector(s->residual_0, residual, N);
    s->hist[s->hist_count++] = s->residual_norm;
    s->stopping = s->o->tol * s->residual_norm;
    do {
        st_precond(s, y, p_old);
        st_spmv(s, y, v);
        double alpha = rho_old / orc_dot(s->residual_0, v, N);
        orc_subtract_vectors(sv, r_old, v, N, alpha);
        st_precond(s, s_tmp, sv);
        st_spmv(s, s_tmp, z);
        double zs = orc_dot(z, sv, N);
        double omega = zs / orc_dot(z, z, N);
        orc_sum_vectors(h, x_old, y, N, alpha);
        orc_sum_vectors(x_new, h, s_tmp, N, omega);
        orc_subtract_vectors(r_new, sv, z, N, omega);
        rho_new = orc_dot(s->residual_0, r_new, N);
        double beta = (rho_new / rho_old) * (alpha / omega);
        orc_subtract_vectors(s->tmp, p_old, v, N, omega);
        orc_sum_vectors(p_new, r_new, s->tmp, N, beta);
        swapd(&residual, &r_new); /* bicgstab.hpp:177 */
        ++s->iter_count;
        s->residual_norm = orc_euclidean_vec_norm(residual, N);
        s->hist[s->hist_count++] = s->residual_norm;
        swapd(&p_old, &p_new);
        swapd(&r_old, &residual);
        swapd(&x_old, &x_new);
        { double t = rho_old; rho_old = rho_new; rho_new = t; }
    } while (!st_stop(s));
    s->residual = residual;
    res->converged = s->residual_norm < s->stopping;
    orc_copy_vector(s->x_star, x_old, N);
    res->final_true_residual = st_final_residual(s);
    free(x_new); free(x_old); free(p_new); free(p_old); free(r_new);
    free(r_old); free(v); free(h); free(sv); free(s_tmp); free(y); free(z);
}

/* ---- Jacobi: methods/jacobi.hpp:43-52, :79-107 ---- */
static void run_jacobi(orc_state *s, orc_result *res) {
    int64_t N = s->N;
    double *x_new = dalloc(N), *x_old = dalloc(N);
    orc_copy_vector(x_old, s->x_0, N);
    orc_compute_residual(&s->A, x_old, s->b, s->residual, s->tmp);
    s->residual_norm = orc_euclidean_vec_norm(s->residual, N);
    orc_copy_vector(s->residual_0, s->residual, N);
    s->hist[s->hist_count++] = s->residual_norm;
    s->stopping = s->o->tol * s->residual_norm;
    do {
        st_spmv(s, x_old, x_new);
        orc_normalize_x(x_new, x_old, s->A_D, s->b, N);
        ++s->iter_count;
        orc_compute_residual(&s->A, x_new, s->b, s->residual, s->tmp);
        s->residual_norm = orc_euclidean_vec_norm(s->residual, N);
        s->hist[s->hist_count++] = s->residual_norm;
        swapd(&x_old, &x_new);
    } while (!st_stop(s));
    res->converged = s->residual_norm < s->stopping;
    orc_copy_vector(s->x_star, x_old, N);
    res->final_true_residual = st_final_residual(s);
    free(x_new); free(x_old);
}

/* ---- GS / SGS: methods/gauss_seidel.hpp:26-52, :76-105, :119-129 ---- */
static void run_gs(orc_state *s, orc_result *res, int symmetric) {
    int64_t N = s->N;
    double *x = dalloc(N);
    orc_copy_vector(x, s->x_0, N);
    orc_compute_residual(&s->A, x, s->b, s->residual, s->tmp);
    s->residual_norm = orc_euclidean_vec_norm(s->residual, N);
    orc_copy_vector(s->residual_0, s->residual, N);
    s->hist[s->hist_count++] = s->residual_norm;
    s->stopping = s->o->tol * s->residual_norm;
    do {
        orc_spmv(N, s->Us.row_ptr, s->Us.col, s->Us.val, x, s->tmp);
        orc_subtract_vectors(s->tmp, s->b, s->tmp, N, 1.0);
        orc_sptrsv(N, s->Ls.row_ptr, s->Ls.col, s->Ls.val, x, s->A_D, s->tmp);
        if (symmetric) {
            orc_spmv(N, s->Ls.row_ptr, s->Ls.col, s->Ls.val, x, s->tmp);
            orc_subtract_vectors(s->tmp, s->b, s->tmp, N, 1.0);
            orc_bsptrsv(N, s->Us.row_ptr, s->Us.col, s->Us.val, x, s->A_D,
                        s->tmp);
        }
        ++s->iter_count;
        orc_compute_residual(&s->A, x, s->b, s->residual, s->tmp);
        s->residual_norm = orc_euclidean_vec_norm(s->residual, N);
        s->hist[s->hist_count++] = s->residual_norm;
    } while (!st_stop(s));
    res->converged = s->residual_norm < s->stopping;
    orc_copy_vector(s->x_star, x, N);
    res->final_true_residual = st_final_residual(s);
    free(x);
}

/* ---- GMRES(m): methods/gmres.hpp.  Small dense algebra follows
 * kernels.hpp:222-310 (row-major, naive triple loops). ---- */
static void dense_identity(double *M, int r, int c) {
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) M[c * i + j] = (i == j) ? 1.0 : 0.0;
}
static void dense_mm(const double *A, const double *B, double *C, int rA,
                     int cA, int cB) { /* dgemm_transpose2, kernels.hpp:273-284 */
    for (int i = 0; i < rA; ++i)
        for (int j = 0; j < cB; ++j) {
            double t = 0.0;
            for (int k = 0; k < cA; ++k) t += A[i * cA + k] * B[k * cB + j];
            C[i * cB + j] = t;
        }
}

typedef struct {
    double *x, *x_old, *V, *Vy, *y, *H, *H_tmp, *J, *Q, *Q_tmp, *w, *R, *g,
        *g_tmp;
    double beta;
    int m;
} gm_t;

/* GMRESSolver::init_structs, gmres.hpp:235-270 (restart part) */
static void gm_reset(orc_state *s, gm_t *g) {
    int m = g->m;
    int64_t N = s->N;
    orc_init_vector(s->tmp, 0.0, N);
    orc_init_vector(s->work, 0.0, N);
    orc_init_vector(s->residual, 0.0, N);
    orc_init_vector(s->residual_0, 0.0, N);
    orc_init_vector(g->V, 0.0, N * (m + 1));
    orc_init_vector(g->Vy, 0.0, N);
    orc_init_vector(g->w, 0.0, N);
    for (int i = 0; i < m; ++i) g->y[i] = 0.0;
    for (int i = 0; i < m + 1; ++i) g->g[i] = g->g_tmp[i] = 0.0;
    for (int i = 0; i < (m + 1) * m; ++i) g->H[i] = g->H_tmp[i] = g->R[i] = 0.0;
    dense_identity(g->J, m + 1, m + 1);
    dense_identity(g->Q, m + 1, m + 1);
    dense_identity(g->Q_tmp, m + 1, m + 1);
}

/* GMRESSolver::init_residual, gmres.hpp:272-318 */
static void gm_init_residual(orc_state *s, gm_t *g, int restarted) {
    int64_t N = s->N;
    orc_compute_residual(&s->A, g->x, s->b, s->residual, s->tmp);
    if (!restarted) {
        s->residual_norm = orc_euclidean_vec_norm(s->residual, N);
        s->hist[s->hist_count++] = s->residual_norm;
    }
    st_precond(s, s->residual, s->residual);
    double pnorm = orc_euclidean_vec_norm(s->residual, N);
    g->beta = pnorm;
    g->g[0] = pnorm;
    g->g_tmp[0] = pnorm;
    orc_scale(g->V, s->residual, 1.0 / pnorm, N);
    if (restarted) {
        s->residual_norm = pnorm;
        orc_copy_vector(s->residual_0, s->residual, N);
        s->hist[s->hist_count++] = s->residual_norm;
    }
}

/* GMRESSolver::get_explicit_x, gmres.hpp:326-375 (y[n]=0 semantics). */
static void gm_explicit_x(orc_state *s, gm_t *g) {
    int m = g->m;
    int n = s->iter_count - s->restart_count * m;
    double diag = 1.0;
    for (int r = n - 1; r >= 0; --r) {
        double sum = 0.0;
        for (int c = r; c < n; ++c) {
            if (r == c) diag = g->R[r * m + c];
            else sum += g->R[r * m + c] * g->y[c];
        }
        g->y[r] = (g->g[r] - sum) / diag;
    }
    orc_multi_axpy(g->V, s->N, g->y, n, g->Vy);
    for (int64_t i = 0; i < s->N; ++i) g->x[i] = g->x_old[i] + g->Vy[i];
}

static void run_gmres(orc_state *s, orc_result *res) {
    int64_t N = s->N;
    int m = s->o->restart_len;
    gm_t g;
    g.m = m;
    g.x = dalloc(N); g.x_old = dalloc(N); g.V = dalloc(N * (m + 1));
    g.Vy = dalloc(N); g.y = dalloc(m); g.H = dalloc((m + 1) * m);
    g.H_tmp = dalloc((m + 1) * m); g.J = dalloc((m + 1) * (m + 1));
    g.Q = dalloc((m + 1) * (m + 1)); g.Q_tmp = dalloc((m + 1) * (m + 1));
    g.w = dalloc(N); g.R = dalloc((m + 1) * m); g.g = dalloc(m + 1);
    g.g_tmp = dalloc(m + 1);
    gm_reset(s, &g);
    orc_copy_vector(g.x, s->x_0, N);
    orc_copy_vector(g.x_old, s->x_0, N);
    gm_init_residual(s, &g, 0);
    s->stopping = s->o->tol * s->residual_norm;
    do {
        int n = s->iter_count - s->restart_count * m;
        /* gmres_separate_iteration, gmres.hpp:150-196 */
        st_spmv(s, &g.V[(int64_t)n * N], g.w);
        st_precond(s, g.w, g.w);
        /* orthogonalize_V (modified Gram-Schmidt), gmres.hpp:6-53 */
        for (int j = 0; j <= n; ++j) {
            double hjn = orc_dot(g.w, &g.V[(int64_t)j * N], N);
            g.H[n + j * m] = hjn;
            orc_subtract_vectors(g.w, g.w, &g.V[(int64_t)j * N], N, hjn);
        }
        double hn1 = orc_euclidean_vec_norm(g.w, N);
        g.H[(n + 1) * m + n] = hn1;
        orc_scale(&g.V[(int64_t)(n + 1) * N], g.w, 1.0 / hn1, N);
        /* least_squares, gmres.hpp:55-121 */
        dense_identity(g.J, m + 1, m + 1);
        dense_identity(g.H_tmp, m + 1, m);
        if (n == 0) memcpy(g.H_tmp, g.H, sizeof(double) * (size_t)((m + 1) * m));
        else dense_mm(g.Q, g.H, g.H_tmp, m + 1, m + 1, m);
        double a = g.H_tmp[n * m + n], bb = g.H_tmp[(n + 1) * m + n];
        double den = sqrt(pow(a, 2) + pow(bb, 2));
        double c_i = a / den, s_i = bb / den;
        g.J[n * (m + 1) + n] = c_i;
        g.J[n * (m + 1) + (n + 1)] = s_i;
        g.J[(n + 1) * (m + 1) + n] = -1.0 * s_i;
        g.J[(n + 1) * (m + 1) + (n + 1)] = c_i;
        dense_mm(g.J, g.Q, g.Q_tmp, m + 1, m + 1, m + 1);
        memcpy(g.Q, g.Q_tmp, sizeof(double) * (size_t)((m + 1) * (m + 1)));
        dense_mm(g.Q, g.H, g.R, m + 1, m + 1, m);
        /* update_g, gmres.hpp:123-148; dgemv kernels.hpp:299-310 */
        for (int i = 0; i < m + 1; ++i) g.g_tmp[i] = 0.0;
        g.g_tmp[0] = g.beta;
        for (int i = 0; i < m + 1; ++i) g.g[i] = g.g_tmp[i];
        for (int i = 0; i < m + 1; ++i) {
            g.g_tmp[i] = 0.0;
            for (int j = 0; j < m + 1; ++j)
                g.g_tmp[i] += 1.0 * g.Q[i * (m + 1) + j] * g.g[j];
        }
        for (int i = 0; i < m + 1; ++i) g.g[i] = g.g_tmp[i];
        s->residual_norm = fabs(g.g[n + 1]);
        ++s->iter_count;
        s->hist[s->hist_count++] = s->residual_norm;
        /* check_restart, gmres.hpp:388-415 */
        int conv = s->residual_norm < s->stopping;
        int over = s->iter_count > s->o->max_iters;
        int cyc = (s->iter_count % m == 0) && s->iter_count != 0;
        if (!conv && !over && cyc) {
            gm_explicit_x(s, &g);
            orc_copy_vector(g.x_old, g.x, N);
            gm_reset(s, &g);
            gm_init_residual(s, &g, 1);
            ++s->restart_count;
        }
    } while (!st_stop(s));
    res->converged = s->residual_norm < s->stopping;
    gm_explicit_x(s, &g);
    orc_copy_vector(s->x_star, g.x, N);
    res->final_true_residual = st_final_residual(s);
    free(g.x); free(g.x_old); free(g.V); free(g.Vy); free(g.y); free(g.H);
    free(g.H_tmp); free(g.J); free(g.Q); free(g.Q_tmp); free(g.w); free(g.R);
    free(g.g); free(g.g_tmp);
}

/* preprocessing (preprocessing.hpp:26-100) -> solve (solver_harness.hpp:7-61)
 * -> summary (postprocessing.hpp:33-68) on a CRS matrix.  `hist` must hold
 * 2*max_iters doubles (solver.hpp:64); x_star N doubles.  The matrix arrays
 * are copied (scaling / diagonal peeling mutate them). */
ORC_API int orc_solve(int64_t n_rows, int64_t nnz, const int64_t *row_ptr,
                      const int32_t *col, const double *val,
                      const orc_opts *o, double *hist, double *x_star,
                      orc_result *res) {
    orc_state s;
    memset(&s, 0, sizeof s);
    s.o = o;
    s.N = n_rows;
    int64_t N = n_rows;
    double *aval = (double *)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1));
    memcpy(aval, val, sizeof(double) * (size_t)nnz);
    s.A.n_rows = s.A.n_cols = N; s.A.nnz = nnz;
    s.A.row_ptr = row_ptr; s.A.col = col; s.A.val = aval;
    s.x_star = x_star;
    s.x_0 = dalloc(N); s.b = dalloc(N); s.tmp = dalloc(N); s.work = dalloc(N);
    s.residual = dalloc(N); s.residual_0 = dalloc(N); s.A_D = dalloc(N);
    s.A_D_inv = dalloc(N); s.A_D_scale = dalloc(N); s.L_D = dalloc(N);
    s.U_D = dalloc(N);
    s.hist = hist;
    for (int i = 0; i < 2 * o->max_iters; ++i) hist[i] = 0.0;
    for (int64_t i = 0; i < N; ++i) { /* solver.hpp:96-108 */
        x_star[i] = 0.0; s.x_0[i] = o->init_x; s.b[i] = o->b_val;
        s.A_D[i] = 1.0; s.L_D[i] = 1.0; s.U_D[i] = 1.0;
    }
    if (o->num_scale) { /* preprocessing.hpp:39-50 */
        int64_t e = orc_extract_scale(&s.A, s.A_D_scale);
        if (e) return 2;
        orc_scale_mat(N, row_ptr, col, aval, s.A_D_scale);
        /* The reference also scales x_0 here (preprocessing.hpp:48), but every
         * solver's init_structs has ALREADY copied the unscaled x_0 into its
         * iterate (preprocessing.hpp:32-33 runs before :39-50), and x_0 is not
         * read again.  So the effective start vector is the unscaled
         * INIT_X_VAL: x_0 is left untouched here. */
        orc_scale_vec(s.b, s.A_D_scale, N);
    }
    /* factor_LU, LU_factors.hpp:900-934 */
    int64_t c4[4];
    orc_split_LU_count(&s.A, c4);
    int64_t *L_rp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N + 1));
    int64_t *Ls_rp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N + 1));
    int64_t *U_rp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N + 1));
    int64_t *Us_rp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N + 1));
    int32_t *L_col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(c4[0] + 1));
    int32_t *Ls_col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(c4[1] + 1));
    int32_t *U_col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(c4[2] + 1));
    int32_t *Us_col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(c4[3] + 1));
    double *L_val = dalloc(c4[0] + 1), *Ls_val = dalloc(c4[1] + 1),
           *U_val = dalloc(c4[2] + 1), *Us_val = dalloc(c4[3] + 1);
    orc_split_LU_fill(&s.A, L_rp, L_col, L_val, Ls_rp, Ls_col, Ls_val, U_rp,
                      U_col, U_val, Us_rp, Us_col, Us_val);
    int rc = 0;
    if (orc_peel_diag_crs(N, L_rp, L_col, L_val, s.A_D, s.A_D_inv) != 0) rc = 3;
    if (!rc && orc_peel_diag_crs(N, U_rp, U_col, U_val, s.A_D, s.A_D_inv) != 0)
        rc = 3;
    if (!rc && o->precond == ORC_PC_ILU0 && o->ilu_real)
        orc_factor_ilu0(&s.A, o->ilu_pivot_tol, o->ilu_pivot_repl, Ls_rp,
                        Ls_col, Ls_val, s.L_D, Us_rp, Us_col, Us_val, s.U_D);
    /* literal reference without SMAX: factor_ILU0_new is a no-op, then
     * peel_diag_crs(U, U_D) gives U_D = diag(A) (LU_factors.hpp:917-920) */
    if (!rc && o->precond == ORC_PC_ILU0 && !o->ilu_real)
        for (int64_t i = 0; i < N; ++i) s.U_D[i] = s.A_D[i];
    s.Ls.n_rows = s.Ls.n_cols = N; s.Ls.nnz = c4[1];
    s.Ls.row_ptr = Ls_rp; s.Ls.col = Ls_col; s.Ls.val = Ls_val;
    s.Us.n_rows = s.Us.n_cols = N; s.Us.nnz = c4[3];
    s.Us.row_ptr = Us_rp; s.Us.col = Us_col; s.Us.val = Us_val;
    if (!rc) {
        switch (o->solver) {
        case ORC_S_CG: run_cg(&s, res); break;
        case ORC_S_BICGSTAB: run_bicgstab(&s, res); break;
        case ORC_S_JACOBI: run_jacobi(&s, res); break;
        case ORC_S_GS: run_gs(&s, res, 0); break;
        case ORC_S_SGS: run_gs(&s, res, 1); break;
        case ORC_S_GMRES: run_gmres(&s, res); break;
        default: rc = 4;
        }
        res->iters = s.iter_count +
                     (o->solver == ORC_S_GMRES ? s.restart_count : 0);
        res->n_hist = s.hist_count;
        res->stopping_criteria = s.stopping;
    }
    free(aval); free(s.x_0); free(s.b); free(s.tmp); free(s.work);
    free(s.residual); free(s.residual_0); free(s.A_D); free(s.A_D_inv);
    free(s.A_D_scale); free(s.L_D); free(s.U_D);
    free(L_rp); free(Ls_rp); free(U_rp); free(Us_rp);
    free(L_col); free(Ls_col); free(U_col); free(Us_col);
    free(L_val); free(Ls_val); free(U_val); free(Us_val);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* cpu_baseline leg of bench.py: the unpreconditioned / Jacobi CG loop of     */
/* methods/cg.hpp:6-54 + :162-166 on a caller-owned CRS, without the setup   */
/* copies of orc_solve (an HPCG-256 matrix is 5.4 GB).  Runs `iters`          */
/* iterations (no stopping test), returns the residual norms in hist[0..iters]*/
/* and the wall time of the iteration loop alone in *loop_seconds.            */
/* ------------------------------------------------------------------------ */
#include <sys/time.h>
static double wall_now(void) {
    struct timeval tv;
    gettimeofday(&tv, 0);
    return tv.tv_sec + 1e-6 * tv.tv_usec;
}
ORC_API int orc_cg_run(int64_t n, const int64_t *row_ptr, const int32_t *col,
                       const double *val, const double *A_D, double b_val,
                       double x0_val, int iters, double *hist,
                       double *loop_seconds) {
    double *x = dalloc(n), *xn = dalloc(n), *b = dalloc(n), *tmp = dalloc(n),
           *p = dalloc(n), *pn = dalloc(n), *r = dalloc(n), *rn = dalloc(n),
           *z = dalloc(n), *zn = dalloc(n);
    orc_init_vector(x, x0_val, n);
    orc_init_vector(b, b_val, n);
    orc_init_vector(xn, 0.0, n); orc_init_vector(tmp, 0.0, n);
    orc_init_vector(p, 0.0, n); orc_init_vector(pn, 0.0, n);
    orc_init_vector(r, 0.0, n); orc_init_vector(rn, 0.0, n);
    orc_init_vector(z, 0.0, n); orc_init_vector(zn, 0.0, n);
    orc_spmv(n, row_ptr, col, val, x, tmp);
    orc_subtract_vectors(r, b, tmp, n, 1.0);
    if (A_D) orc_elemwise_div_vectors(z, r, A_D, n, 1.0); else orc_copy_vector(z, r, n);
    orc_copy_vector(p, z, n);
    hist[0] = orc_euclidean_vec_norm(r, n);
    double t0 = wall_now();
    for (int it = 0; it < iters; ++it) {
        orc_spmv(n, row_ptr, col, val, p, tmp);
        double tmp_dot = orc_dot(r, z, n);
        double alpha = tmp_dot / orc_dot(tmp, p, n);
        orc_sum_vectors(xn, x, p, n, alpha);
        orc_subtract_vectors(rn, r, tmp, n, alpha);
        if (A_D) orc_elemwise_div_vectors(zn, rn, A_D, n, 1.0); else orc_copy_vector(zn, rn, n);
        double beta = orc_dot(rn, zn, n) / tmp_dot;
        orc_sum_vectors(pn, zn, p, n, beta);
        hist[it + 1] = orc_euclidean_vec_norm(rn, n);
        swapd(&p, &pn); swapd(&z, &zn); swapd(&r, &rn); swapd(&x, &xn);
    }
    *loop_seconds = wall_now() - t0;
    free(x); free(xn); free(b); free(tmp); free(p); free(pn); free(r); free(rn);
    free(z); free(zn);
    return 0;
}

/* ---- host-memory probe for bench.py's cpu_baseline.topology (no reference counterpart: it explains the CPU leg) ----
 * STREAM-triad-shaped loop a[i] = b[i] + s c[i] -- the shape of the reference's sum_vectors (kernels.hpp:128-135) -- over
 * vectors of n doubles that the SAME static OpenMP schedule touched first (the placement MatrixCRS::operator= gives the
 * reference's arrays, sparse_matrix.hpp:92-128).  Returns GB/s (24 n bytes per pass), the best of `reps` passes.
 * places_out (may be NULL): the CPU each of the first `cap` threads ran on, by sched_getcpu(). */
extern int sched_getcpu(void); /* glibc; declared here so that the file needs no _GNU_SOURCE */
ORC_API double orc_host_triad(int64_t n, int reps, int *places_out, int cap) {
    double *a = dalloc(n), *b = dalloc(n), *c = dalloc(n);
    if (!a || !b || !c) { free(a); free(b); free(c); return -1.0; }
#pragma omp parallel
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
        if (places_out && t < cap) places_out[t] = sched_getcpu();
#endif
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) { a[i] = 0.0; b[i] = 1.0; c[i] = 2.0; }
    }
    double best = 0.0;
    for (int r = 0; r < reps; ++r) {
        const double t0 = wall_now();
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) a[i] = b[i] + 0.5 * c[i];
        const double dt = wall_now() - t0;
        const double gbs = 24.0 * (double)n / dt / 1e9;
        if (gbs > best) best = gbs;
    }
    volatile double sink = a[n / 2]; (void)sink;
    free(a); free(b); free(c);
    return best;
}

/* ---- MatrixMarket writer for the ingestion figure (tools/mtx_ingest.py; test infrastructure) ----------------------------
 * Writes a CRS matrix as `%%MatrixMarket matrix coordinate real general|symmetric`.  general: row by row, the entries of a row in
 * CRS order (the reference's reader keeps the file's order inside a row, sparse_matrix.hpp:332-344).  symmetric: the LOWER
 * triangle in column-major order -- the layout of the SuiteSparse files, from which the reference's reader (mirrored entry right
 * behind its source entry, stable sort by row: :308-318) rebuilds rows with ascending columns; taken from the rows of the upper
 * triangle, so the matrix must be symmetric in pattern and values (the caller checks).  Values with %.17g (round-trip exact).
 * Each OpenMP thread formats a run of rows into its own buffer; the buffers are written in order.  Returns entries written. */
ORC_API int64_t orc_write_mtx(const char *path, int64_t n, const int64_t *row_ptr, const int32_t *col, const double *val, int symmetric) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    int64_t stored = 0;
    if (symmetric) { for (int64_t r = 0; r < n; ++r) for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) stored += col[k] >= r; }
    else stored = row_ptr[n];
    fprintf(f, "%%%%MatrixMarket matrix coordinate real %s\n%% written by oracle/bis_oracle.c orc_write_mtx\n%lld %lld %lld\n",
            symmetric ? "symmetric" : "general", (long long)n, (long long)n, (long long)stored);
    int T = 1;
#ifdef _OPENMP
    T = omp_get_max_threads();
#endif
    const int64_t pieces = (int64_t)T * 8;
    int64_t ok = 1;
    for (int64_t p0 = 0; p0 < pieces && ok; p0 += T) { /* T pieces at a time: bounded memory */
        char *bufs[1024]; size_t lens[1024];
        const int np = (int)((pieces - p0) < T ? (pieces - p0) : T);
#pragma omp parallel for schedule(static, 1)
        for (int t = 0; t < np; ++t) {
            const int64_t a = n * (p0 + t) / pieces, b = n * (p0 + t + 1) / pieces;
            const size_t cap = (size_t)(row_ptr[b] - row_ptr[a]) * 48 + 64;
            char *buf = (char *)malloc(cap);
            size_t len = 0;
            if (buf)
                for (int64_t r = a; r < b; ++r)
                    for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
                        if (symmetric && col[k] < r) continue;
                        if (symmetric) len += (size_t)snprintf(buf + len, 48, "%d %lld %.17g\n", col[k] + 1, (long long)r + 1, val[k]);
                        else len += (size_t)snprintf(buf + len, 48, "%lld %d %.17g\n", (long long)r + 1, col[k] + 1, val[k]);
                    }
            bufs[t] = buf; lens[t] = len;
        }
        for (int t = 0; t < np; ++t) {
            if (!bufs[t] || fwrite(bufs[t], 1, lens[t], f) != lens[t]) ok = 0;
            free(bufs[t]);
        }
    }
    if (fclose(f) != 0) ok = 0;
    return ok ? stored : -1;
}
