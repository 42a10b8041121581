// bis_stationary.hip -- the stationary solvers of the reference (methods/jacobi.hpp:43-52, :102-107;
// methods/gauss_seidel.hpp:26-52, :99-104) as device schedules: the iteration, the true residual
// b - A x of every iterate, its norm and the stopping test of solver.hpp:177-192 all run on the
// device; the host enqueues iterations and reads the status when it wants to.  Once the stop test
// fires every later launch is a no-op (SpMVs, sweeps and the passes here all look at the stop flag),
// so x is exactly the reference's iterate at its stopping iteration.
//
// Jacobi.  The reference's iteration k makes x_k from tmp = A x_{k-1} (spmv + normalize_x) and then
// samples the true residual with a second product A x_k (compute_residual) -- which is the very
// vector iteration k+1 starts from.  The schedule here keeps it: one SpMV per iteration,
//     t_k = A x_k ;  r_k = b - t_k, ||r_k||  (the sample of iteration k) ;  x_{k+1} = (b - (t_k - D x_k)) / D
// with the last two in ONE pass over t_k (32 N bytes read, 8 N written).  Per element and per
// partial sum the arithmetic is that of the separate kernels (ew3 OP_SUB, dot_partial_kernel's index
// map and accumulators, reduce_finish, normalize_x_kernel): the residual history is bit-identical
// to the kernel-by-kernel schedule.  x_{k+1} goes to the other of two buffers, so the iterate the
// stop test accepts is still intact when it fires.
//
// Gauss-Seidel / symmetric Gauss-Seidel.  The reference's operations unchanged (tmp = U x; tmp = b -
// tmp; x = (D+L)^-1 tmp; [the mirrored backward sweep]; tmp = A x; r = b - tmp; ||r||), stream-ordered,
// with the norm and the stop test on the device instead of a blocking read per iteration.
#include "bis_internal.hpp"

#include <cfloat>
#include <cmath>

struct bis_stat {
    int kind = 0; // BIS_STAT_JACOBI / _GS / _SGS
    const bis_mat *A = nullptr, *L = nullptr, *U = nullptr;
    const double *D = nullptr, *b = nullptr;
    double *x = nullptr;   // the caller's iterate (Jacobi: buffer 0)
    double *xb = nullptr;  // Jacobi: buffer 1
    double *t = nullptr;   // A x
    double *r = nullptr;   // GS: b - A x
    int64_t n = 0;
    double *sc = nullptr;  // device: [0] stopping threshold, [1] sum of squares of the current sample
    int *flags = nullptr;  // device: [0] iterations done, [1] stopped, [2] converged, [3] stopping iteration
    double *hist = nullptr;
    int hist_cap = 1 << 16;
    int enqueued = 0;
    bool initialised = false;
};

namespace {

constexpr int kT = 256;

inline int ew_grid(int64_t n_items) {
    int64_t g = (n_items + kT - 1) / kT;
    if (g < 1) g = 1;
    if (g > kMaxDotBlocks) g = kMaxDotBlocks; // (the index map of dot_partial_kernel: bis_blas1.hip)
    return (int)g;
}

// Jacobi: r = b - t (kernels.hpp:124 with scale 1: fma(-1, t, b)), partials of (r, r) with the index map and the two
// accumulators of dot_partial_kernel, and x_next = (b - (t - D x)) / D as normalize_x_kernel computes it.
template <bool VEC>
__global__ __launch_bounds__(kT) void jacobi_step_kernel(int64_t n, const int *__restrict__ flags, const double *__restrict__ t,
                                                         const double *__restrict__ D, const double *__restrict__ b,
                                                         const double *__restrict__ x, double *__restrict__ x_next,
                                                         double *__restrict__ partials) {
    __shared__ double lds[kT / 64];
    if (flags[1]) return;
    const int64_t stride = (int64_t)gridDim.x * kT;
    int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    double acc0 = 0.0, acc1 = 0.0;
    auto next = [](double tv, double dv, double xv, double bv) {
        const double adjusted = fma(-dv, xv, tv);
        return (bv - adjusted) / dv;
    };
    if (VEC) {
        const int64_t n2 = n >> 1;
        const double2 *t2 = reinterpret_cast<const double2 *>(t), *D2 = reinterpret_cast<const double2 *>(D);
        const double2 *b2 = reinterpret_cast<const double2 *>(b), *x2 = reinterpret_cast<const double2 *>(x);
        double2 *o2 = reinterpret_cast<double2 *>(x_next);
        for (; i < n2; i += stride) {
            const double2 tv = t2[i], bv = b2[i], dv = D2[i], xv = x2[i];
            const double r0 = fma(-1.0, tv.x, bv.x), r1 = fma(-1.0, tv.y, bv.y);
            acc0 = fma(r0, r0, acc0);
            acc1 = fma(r1, r1, acc1);
            double2 o;
            o.x = next(tv.x, dv.x, xv.x, bv.x);
            o.y = next(tv.y, dv.y, xv.y, bv.y);
            o2[i] = o;
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const double rr = fma(-1.0, t[n - 1], b[n - 1]);
            acc0 = fma(rr, rr, acc0);
            x_next[n - 1] = next(t[n - 1], D[n - 1], x[n - 1], b[n - 1]);
        }
    } else {
        for (; i < n; i += stride) {
            const double rr = fma(-1.0, t[i], b[i]);
            acc0 = fma(rr, rr, acc0);
            x_next[i] = next(t[i], D[i], x[i], b[i]);
        }
    }
    const double s = block_sum<kT>(acc0 + acc1, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// sum of the partials in reduce_finish_kernel's order, then the bookkeeping of one iteration: the sampled norm
// (kernels.hpp:202), the iteration count (solver_harness.hpp:21) and check_stopping_criteria (solver.hpp:177-192;
// max_iters is the host's).  partials == nullptr: the sum of squares is already in sc[1].
__global__ __launch_bounds__(256) void stat_book_kernel(const double *__restrict__ partials, int n_partials, double *sc, int *flags,
                                                        double *hist, int hist_cap) {
    __shared__ double lds[4];
    if (flags[1]) return;
    double ss;
    if (partials) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < n_partials; i += 256) acc += partials[i];
        ss = block_sum<256>(acc, lds);
    } else {
        ss = sc[1];
    }
    if (threadIdx.x != 0) return;
    const double norm = sqrt(ss);
    const int it = flags[0] + 1;
    flags[0] = it;
    if (it < hist_cap) hist[it] = norm;
    const bool conv = fabs(norm) < sc[0];
    const bool diverged = fabs(norm) > DBL_MAX || norm != norm;
    if (conv || diverged) { flags[1] = 1; flags[2] = conv ? 1 : 0; flags[3] = it; }
}

} // namespace

extern "C" {

bis_status bis_stat_destroy(bis_ctx *ctx, bis_stat *s) {
    BIS_CTX_OK(ctx);
    if (!s) return BIS_OK;
    hipStreamSynchronize(ctx->stream);
    hipFree(s->xb); hipFree(s->t); hipFree(s->r); hipFree(s->sc); hipFree(s->flags); hipFree(s->hist);
    delete s;
    return BIS_OK;
}

bis_status bis_stat_create(bis_ctx *ctx, int kind, const bis_mat *A, const bis_mat *L_strict, const bis_mat *U_strict,
                           const double *D, const double *b, double *x, bis_stat **out) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, out && A && D && b && x && kind >= BIS_STAT_JACOBI && kind <= BIS_STAT_SGS, "bis_stat_create: bad arguments");
    BIS_REQUIRE(ctx, A->n_rows == A->n_cols, "bis_stat_create: square matrix required");
    BIS_REQUIRE(ctx, kind == BIS_STAT_JACOBI || (L_strict && U_strict && L_strict->n_rows == A->n_rows && U_strict->n_rows == A->n_rows),
                "bis_stat_create: the sweeps need the strict triangles of A");
    bis_stat *s = new bis_stat;
    s->kind = kind;
    s->A = A; s->L = L_strict; s->U = U_strict;
    s->D = D; s->b = b; s->x = x;
    s->n = A->n_rows;
    bis_status st = bis_vec_alloc(ctx, s->n, &s->t);
    if (st == BIS_OK && kind == BIS_STAT_JACOBI) st = bis_vec_alloc(ctx, s->n, &s->xb);
    if (st == BIS_OK && kind != BIS_STAT_JACOBI) st = bis_vec_alloc(ctx, s->n, &s->r);
    if (st == BIS_OK) st = bis_vec_alloc(ctx, 8, &s->sc);
    if (st == BIS_OK) st = bis_vec_alloc(ctx, s->hist_cap, &s->hist);
    if (st == BIS_OK && hipMalloc(&s->flags, sizeof(int) * 4) != hipSuccess) st = BIS_ERR_HIP;
    if (st != BIS_OK) { bis_stat_destroy(ctx, s); return st; }
    *out = s;
    return BIS_OK;
}

// init_residual (jacobi.hpp:79-84, gauss_seidel.hpp:76-81): r_0 = b - A x_0, its norm, the stopping threshold
bis_status bis_stat_init(bis_ctx *ctx, bis_stat *s, double tol, double *r0_norm_host) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, s, "bis_stat_init: null handle");
    const int64_t n = s->n;
    bis_status st = bis_ensure_partials(ctx, (size_t)2 * kMaxDotBlocks);
    if (st != BIS_OK) return st;
    double *r = s->kind == BIS_STAT_JACOBI ? s->xb : s->r; // (Jacobi: buffer 1 is free until the first step)
    st = bis_compute_residual(ctx, s->A, s->x, s->b, r, s->t); // leaves t = A x_0: the first Jacobi step starts from it
    double rr = 0.0;
    if (st == BIS_OK) st = bis_dot(ctx, r, r, n, &rr);
    if (st != BIS_OK) return st;
    const double norm0 = sqrt(rr);
    const double sc[8] = {tol * norm0, 0, 0, 0, 0, 0, 0, 0}; // init_stopping_criteria, solver.hpp:173-175
    const int flags[4] = {0, 0, 0, 0};
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(s->sc, sc, sizeof sc, hipMemcpyHostToDevice, ctx->stream));
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(s->flags, flags, sizeof flags, hipMemcpyHostToDevice, ctx->stream));
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(s->hist, &norm0, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    s->enqueued = 0;
    s->initialised = true;
    if (r0_norm_host) *r0_norm_host = norm0;
    return BIS_OK;
}

bis_status bis_stat_iterate(bis_ctx *ctx, bis_stat *s, int n_iters) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, s && s->initialised && n_iters >= 0 && s->enqueued + n_iters < s->hist_cap, "bis_stat_iterate: bad arguments");
    const int64_t n = s->n;
    if (n == 0) return BIS_OK;
    ctx->spmv_stop = s->flags;
    struct StopGuard { bis_ctx *c; ~StopGuard() { c->spmv_stop = nullptr; } } stop_guard{ctx};
    bis_status st = BIS_OK;
    int done = 0;
    for (; done < n_iters && st == BIS_OK; ++done) {
        const int k = s->enqueued + done; // iterations completed before this one
        if (s->kind == BIS_STAT_JACOBI) {
            // t = A x_k is at hand (from bis_stat_init or from the previous trip); x_{k+1} and the SpMV of the new iterate
            double *cur = (k & 1) ? s->xb : s->x, *nxt = (k & 1) ? s->x : s->xb;
            if (k == 0) { // the norm of r_0 is known: only the step x_1 = (b - (t_0 - D x_0)) / D
                st = bis_copy_vector(ctx, nxt, s->t, n);
                if (st == BIS_OK) st = bis_normalize_x(ctx, nxt, cur, s->D, s->b, n);
            }
            if (st == BIS_OK) st = bis_spmv_launch(ctx, s->A, nxt, s->t, nullptr, nullptr);
            if (st != BIS_OK) break;
            // the sample of this iteration (||b - A x_{k+1}||) and, in the same pass, the step the NEXT iteration starts with
            double *nn = (k & 1) ? s->xb : s->x; // x_{k+2} overwrites x_k
            const bool vec = n >= 2 && (((uintptr_t)s->t | (uintptr_t)s->D | (uintptr_t)s->b | (uintptr_t)nxt | (uintptr_t)nn) & 15) == 0;
            const int grid = vec ? ew_grid(n >> 1) : ew_grid(n);
            if (vec) hipLaunchKernelGGL((jacobi_step_kernel<true>), dim3(grid), dim3(kT), 0, ctx->stream, n, s->flags, s->t, s->D, s->b, nxt, nn, ctx->partials);
            else hipLaunchKernelGGL((jacobi_step_kernel<false>), dim3(grid), dim3(kT), 0, ctx->stream, n, s->flags, s->t, s->D, s->b, nxt, nn, ctx->partials);
            hipLaunchKernelGGL(stat_book_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->partials, grid, s->sc, s->flags, s->hist, s->hist_cap);
        } else {
            // gs_separate_iteration, gauss_seidel.hpp:26-38 (tmp <- U x; tmp <- b - tmp; x <- (D+L)^-1 tmp)
            st = bis_spmv_launch(ctx, s->U, s->x, s->t, nullptr, nullptr);
            if (st == BIS_OK) st = bis_subtract_vectors(ctx, s->t, s->b, s->t, n, 1.0);
            if (st == BIS_OK) st = bis_sptrsv(ctx, s->L, s->x, s->D, s->t);
            if (st == BIS_OK && s->kind == BIS_STAT_SGS) { // bgs_separate_iteration, :40-52
                st = bis_spmv_launch(ctx, s->L, s->x, s->t, nullptr, nullptr);
                if (st == BIS_OK) st = bis_subtract_vectors(ctx, s->t, s->b, s->t, n, 1.0);
                if (st == BIS_OK) st = bis_bsptrsv(ctx, s->U, s->x, s->D, s->t);
            }
            // record_residual_norm, gauss_seidel.hpp:99-104
            if (st == BIS_OK) st = bis_spmv_launch(ctx, s->A, s->x, s->t, nullptr, nullptr);
            if (st == BIS_OK) st = bis_subtract_vectors(ctx, s->r, s->b, s->t, n, 1.0);
            if (st == BIS_OK) st = bis_dot_dev(ctx, s->r, s->r, n, s->sc + 1);
            if (st != BIS_OK) break;
            hipLaunchKernelGGL(stat_book_kernel, dim3(1), dim3(256), 0, ctx->stream, (const double *)nullptr, 0, s->sc, s->flags, s->hist, s->hist_cap);
        }
    }
    s->enqueued += done;
    if (st != BIS_OK) return st;
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}

bis_status bis_stat_status(bis_ctx *ctx, bis_stat *s, int *iters, int *converged, double *hist_host, int hist_cap) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, s, "bis_stat_status: null handle");
    int flags[4] = {0, 0, 0, 0};
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(flags, s->flags, sizeof flags, hipMemcpyDeviceToHost, ctx->stream));
    BIS_SYNC_CHECK(ctx);
    if (iters) *iters = flags[0];
    if (converged) *converged = flags[2];
    if (hist_host && hist_cap > 0) {
        int cnt = flags[0] + 1;
        if (cnt > hist_cap) cnt = hist_cap;
        if (cnt > s->hist_cap) cnt = s->hist_cap;
        BIS_HIP_CHECK(ctx, hipMemcpyAsync(hist_host, s->hist, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream));
        BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return BIS_OK;
}

// the iterate the history's last entry belongs to, copied to x_out (may be the x given to bis_stat_create)
bis_status bis_stat_solution(bis_ctx *ctx, bis_stat *s, double *x_out) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, s && x_out, "bis_stat_solution: bad arguments");
    int it = 0;
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&it, s->flags, sizeof it, hipMemcpyDeviceToHost, ctx->stream));
    BIS_SYNC_CHECK(ctx);
    const double *src = s->x;
    if (s->kind == BIS_STAT_JACOBI && (it & 1)) src = s->xb; // x_k lives in buffer k mod 2
    if (src != x_out) return bis_copy_vector(ctx, x_out, src, s->n);
    return BIS_OK;
}

} // extern "C"
