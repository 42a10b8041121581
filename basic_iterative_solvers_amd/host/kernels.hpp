// kernels.hpp -- the reference's operator surface (kernels.hpp:22-414,
// methods/jacobi.hpp:27-40) with the same names, argument order and default
// arguments, implemented by forwarding to the gfx950 C-ABI library.  Vector
// arguments are device pointers; `dot` / `euclidean_vec_norm` return the
// scalar to the host like the reference (a stream sync).  Aliased operands are
// legal exactly where the reference's callers alias them.
#pragma once

#include "common.hpp"
#include "sparse_matrix.hpp"

// The reference brackets its three sparse kernels with LIKWID markers ("spmv", "sptrsv",
// "backwards-sptrsv": kernels.hpp:25-39, :56-74, :90-105); here the same names are roctx ranges
// (rocprofv3 --marker-trace), around the enqueue of the device kernels.
#ifdef USE_ROCTX
#include <roctracer/roctx.h>
struct MarkerRange { explicit MarkerRange(const char *n) { roctxRangePushA(n); } ~MarkerRange() { roctxRangePop(); } };
#else
struct MarkerRange { explicit MarkerRange(const char *) {} };
#endif

inline void spmv(const MatrixCRS *A, const double *x, double *y, int = 0, Interface * = nullptr,
                 const std::string = "") {
    MarkerRange m("spmv");
    bis::check(bis_spmv(bis::ctx(), A->dev, x, y), "spmv");
}
inline void sptrsv(const MatrixCRS *L, double *x, const double *D, const double *b, int = 0,
                   Interface * = nullptr, const std::string = "") {
    MarkerRange m("sptrsv");
    bis::check(bis_sptrsv(bis::ctx(), L->dev, x, D, b), "sptrsv");
}
inline void bsptrsv(const MatrixCRS *U, double *x, const double *D, const double *b, int = 0,
                    Interface * = nullptr, const std::string = "") {
    MarkerRange m("backwards-sptrsv");
    bis::check(bis_bsptrsv(bis::ctx(), U->dev, x, D, b), "bsptrsv");
}
inline void subtract_vectors(double *r, const double *a, const double *b, const int N, const double scale = 1.0) {
    bis::check(bis_subtract_vectors(bis::ctx(), r, a, b, N, scale), "subtract_vectors");
}
inline void sum_vectors(double *r, const double *a, const double *b, const int N, const double scale = 1.0) {
    bis::check(bis_sum_vectors(bis::ctx(), r, a, b, N, scale), "sum_vectors");
}
inline void elemwise_mult_vectors(double *r, const double *a, const double *b, const int N, const double scale = 1.0) {
    bis::check(bis_elemwise_mult_vectors(bis::ctx(), r, a, b, N, scale), "elemwise_mult_vectors");
}
inline void elemwise_div_vectors(double *r, const double *a, const double *b, const int N, const double scale = 1.0) {
    bis::check(bis_elemwise_div_vectors(bis::ctx(), r, a, b, N, scale), "elemwise_div_vectors");
}
inline void compute_residual(const MatrixCRS *A, const double *x, const double *b, double *residual,
                             double *tmp, Interface * = nullptr, const std::string = "") {
    bis::check(bis_compute_residual(bis::ctx(), A->dev, x, b, residual, tmp), "compute_residual");
}
inline double euclidean_vec_norm(const double *v, int N) {
    double r = 0.0;
    bis::check(bis_euclidean_vec_norm(bis::ctx(), v, N, &r), "euclidean_vec_norm");
    return r;
}
inline double dot(const double *a, const double *b, const int N) {
    double r = 0.0;
    bis::check(bis_dot(bis::ctx(), a, b, N, &r), "dot");
    return r;
}
inline void scale(double *r, const double *v, const double scalar, const int N) {
    bis::check(bis_scale(bis::ctx(), r, v, scalar, N), "scale");
}
inline void init_vector(double *v, double val, long size) {
    bis::check(bis_init_vector(bis::ctx(), v, val, size), "init_vector");
}
inline void copy_vector(double *out, const double *in, const int n) {
    bis::check(bis_copy_vector(bis::ctx(), out, in, n), "copy_vector");
}
inline void normalize_x(double *x_new, const double *x_old, const double *D, const double *b, const int n) {
    bis::check(bis_normalize_x(bis::ctx(), x_new, x_old, D, b, n), "normalize_x");
}
// Vy = sum_{k<n_vec} y[k] V_k  (dgemm_transpose1 as gmres.hpp:358 uses it; y on the host)
inline void multi_axpy(const double *V, const double *y_host, double *out, int N, int n_vec) {
    bis::check(bis_multi_axpy(bis::ctx(), V, N, y_host, n_vec, out, N), "multi_axpy");
}
inline void two_stage_gauss_seidel(const MatrixCRS *strict, double *tmp, double *work, double *D_inv,
                                   double *input, double *output, const int N, int = 0,
                                   Interface * = nullptr, const std::string & = "") {
    bis::check(bis_two_stage_gauss_seidel(bis::ctx(), strict->dev, tmp, work, D_inv, input, output, N,
                                          PRECOND_INNER_ITERS), "two_stage_gauss_seidel");
}
inline void apply_preconditioner(const PrecondType pc, const int N, const MatrixCRS *L_strict,
                                 const MatrixCRS *U_strict, double *A_D, double *A_D_inv, double *L_D,
                                 double *U_D, double *output, double *input, double *tmp, double *work,
                                 int = 0, Interface * = nullptr, const std::string = "") {
    bis::check(bis_apply_preconditioner(bis::ctx(), static_cast<int>(pc), N,
                                        L_strict ? L_strict->dev : nullptr,
                                        U_strict ? U_strict->dev : nullptr, A_D, A_D_inv, L_D, U_D,
                                        output, input, tmp, work, PRECOND_OUTER_ITERS,
                                        PRECOND_INNER_ITERS), "apply_preconditioner");
}

// ---- device-scalar forms (this build's addition): the factor / the result lives in device memory, no host round trip.
// Same kernels and the same IEEE operations in the same order as the host-scalar functions above.
inline void dot_dev(const double *a, const double *b, const int N, double *result_dev) {
    bis::check(bis_dot_dev(bis::ctx(), a, b, N, result_dev), "dot_dev");
}
inline void axpy_dot_dev(double *w, const double *u, const double *scale_dev, const double *v, const int N, double *result_dev) {
    bis::check(bis_axpy_dot_dev(bis::ctx(), w, u, scale_dev, v, N, result_dev), "axpy_dot_dev");
}
inline void subtract_vectors_dev(double *r, const double *a, const double *b, const int N, const double *scale_dev) {
    bis::check(bis_subtract_vectors_dev(bis::ctx(), r, a, b, N, scale_dev), "subtract_vectors_dev");
}
inline void sum_vectors_dev(double *r, const double *a, const double *b, const int N, const double *scale_dev) {
    bis::check(bis_sum_vectors_dev(bis::ctx(), r, a, b, N, scale_dev), "sum_vectors_dev");
}
inline void scale_dev(double *r, const double *v, const double *scalar_dev, const int N) {
    bis::check(bis_scale_dev(bis::ctx(), r, v, scalar_dev, N), "scale_dev");
}
inline void scalar_div(double *out, const double *a, const double *b) { bis::check(bis_scalar_div(bis::ctx(), out, a, b), "scalar_div"); }
inline void scalar_ratio_product(double *out, const double *a, const double *b, const double *c, const double *d) {
    bis::check(bis_scalar_ratio_product(bis::ctx(), out, a, b, c, d), "scalar_ratio_product");
}
inline void scalar_sqrt_inv(double *norm, double *inv, const double *sumsq) {
    bis::check(bis_scalar_sqrt_inv(bis::ctx(), norm, inv, sumsq), "scalar_sqrt_inv");
}

// ---- small dense helpers of GMRES: stay on the host (<= 51x51), kernels.hpp:222-310
inline void init_dense_identity_matrix(double *m, int r, int c) {
    for (int i = 0; i < r; ++i) for (int j = 0; j < c; ++j) m[c * i + j] = i == j ? 1.0 : 0.0;
}
inline void copy_dense_matrix(double *dst, const double *src, int r, int c) { std::copy(src, src + (size_t)r * c, dst); }
inline void dgemm_transpose2(const double *A, const double *B, double *C, int rA, int cA, int cB) {
    for (int i = 0; i < rA; ++i)
        for (int j = 0; j < cB; ++j) {
            double t = 0.0;
            for (int k = 0; k < cA; ++k) t += A[i * cA + k] * B[k * cB + j];
            C[i * cB + j] = t;
        }
}
inline void dgemv(const double *A, const double *x, double *y, int r, int c, double alpha = 1.0) {
    for (int i = 0; i < r; ++i) {
        y[i] = 0.0;
        for (int j = 0; j < c; ++j) y[i] += alpha * A[i * c + j] * x[j];
    }
}
