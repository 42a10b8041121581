// bis_ilu0.hip -- ILU(0) factorisation on the device.
//
// Semantics: the reference's serial factor_ILU0_old (utilities/LU_factors.hpp:
// 320-539) -- row-wise IKJ elimination restricted to A's pattern, dependencies
// k < i taken in ascending column order, |u_kk| < 1e-16 skips a step, an update
// touches only positions whose current value is non-zero (:383), |u_ii| <
// pivot_tol is replaced by sign(u_ii)*pivot_repl (:410-412); outputs L_strict
// and U_strict with ascending columns, L_D = 1, U_D = diag(U).
//
// Parallel form: row i needs the finished rows k < i of its pattern, i.e. the
// dependency levels of A's strict lower triangle (the level-parallel variant
// the reference keeps behind SMAX, LU_factors.hpp:541-768).  One launch per
// level, one lane per row; rows of a level are independent.  The per-row
// arithmetic is sequential and identical to the serial algorithm, so the
// factors do not depend on the schedule.
#include "bis_internal.hpp"

#include <algorithm>

namespace {

// W = A with ascending columns inside each row (stable for duplicate columns);
// also diag position (-1 if absent) and the first position with col > row.
template <typename RP>
__global__ __launch_bounds__(256) void sort_rows_kernel(const RP *__restrict__ rp,
                                                        const int32_t *__restrict__ col,
                                                        const double *__restrict__ val, int64_t n,
                                                        int32_t *__restrict__ wcol,
                                                        double *__restrict__ wval,
                                                        int64_t *__restrict__ dpos,
                                                        int64_t *__restrict__ ustart) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t s = rp[i], e = rp[i + 1];
    for (int64_t p = s; p < e; ++p) { // rank sort, O(len^2), rows are short
        const int32_t c = col[p];
        int64_t rank = 0;
        for (int64_t q = s; q < e; ++q) rank += (col[q] < c) || (col[q] == c && q < p);
        wcol[s + rank] = c;
        wval[s + rank] = val[p];
    }
    int64_t d = -1, u = e;
    for (int64_t p = s; p < e; ++p) { // on the unsorted row: count instead of search
        if (col[p] == i) d = 0; // marker, resolved below
    }
    int64_t less = 0, leq = 0;
    for (int64_t p = s; p < e; ++p) { less += col[p] < i; leq += col[p] <= i; }
    if (d == 0) d = s + less; // first entry equal to i in the sorted row
    u = s + leq;
    dpos[i] = d;
    ustart[i] = u;
}

template <typename RP>
__global__ __launch_bounds__(256) void ilu0_level_kernel(const RP *__restrict__ rp,
                                                         const int32_t *__restrict__ wcol, double *wval,
                                                         const int64_t *__restrict__ dpos,
                                                         const int64_t *__restrict__ ustart,
                                                         const int32_t *__restrict__ perm, int64_t begin,
                                                         int64_t end, double pivot_tol, double pivot_repl,
                                                         double *U_D, double *L_D) {
    const int64_t t = begin + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= end) return;
    const int i = perm[t];
    const int64_t s = rp[i], e = rp[i + 1];
    for (int64_t p = s; p < e; ++p) {
        const int k = wcol[p];
        if (k >= i) break;
        const double pivot = U_D[k]; // row k finished in an earlier level
        if (fabs(pivot) < 1e-16) continue;
        const double factor = wval[p] / pivot;
        wval[p] = factor;
        for (int64_t q = ustart[k]; q < (int64_t)rp[k + 1]; ++q) {
            const int j = wcol[q];
            int64_t lo = p + 1, hi = e; // j > k: search the rest of row i
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (wcol[mid] < j) lo = mid + 1; else hi = mid;
            }
            if (lo < e && wcol[lo] == j && wval[lo] != 0.0) wval[lo] = fma(-factor, wval[q], wval[lo]);
        }
    }
    double u_diag = dpos[i] >= 0 ? wval[dpos[i]] : 0.0;
    if (fabs(u_diag) < pivot_tol) u_diag = (u_diag >= 0 ? 1.0 : -1.0) * pivot_repl;
    if (dpos[i] >= 0) wval[dpos[i]] = u_diag;
    U_D[i] = u_diag;
    L_D[i] = 1.0;
}

template <typename RP>
bis_status ilu0_t(bis_ctx *ctx, const bis_mat *A, double pivot_tol, double pivot_repl, bis_mat **Ls_out,
                  bis_mat **Us_out, double *L_D, double *U_D) {
    const int64_t n = A->n_rows;
    bis_mat *W = nullptr;
    bis_status st = bis_mat_alloc(ctx, n, A->n_cols, A->nnz, A->rp64, &W);
    if (st != BIS_OK) return st;
    int64_t *dpos = nullptr, *ustart = nullptr;
    auto cleanup = [&](bis_status rc) {
        hipFree(dpos);
        hipFree(ustart);
        bis_mat_destroy(ctx, W);
        return rc;
    };
    hipError_t e = hipMalloc(&dpos, sizeof(int64_t) * (size_t)std::max<int64_t>(n, 1));
    if (e == hipSuccess) e = hipMalloc(&ustart, sizeof(int64_t) * (size_t)std::max<int64_t>(n, 1));
    if (e == hipSuccess)
        e = hipMemcpyAsync(W->row_ptr, A->row_ptr, sizeof(RP) * (size_t)(n + 1), hipMemcpyDeviceToDevice,
                           ctx->stream);
    if (e != hipSuccess) { ctx->err = std::string("bis_mat_ilu0: ") + hipGetErrorString(e); return cleanup(BIS_ERR_HIP); }
    const RP *rp = (const RP *)W->row_ptr;
    if (n > 0)
        hipLaunchKernelGGL(sort_rows_kernel<RP>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const RP *)A->row_ptr, A->col, A->val, n, W->col, W->val, dpos, ustart);
    st = bis_mat_finalize(ctx, W);
    if (st != BIS_OK) return cleanup(st);
    // dependency levels = levels of the strict lower triangle of the pattern
    bis_mat *Lp = nullptr, *Up = nullptr;
    st = bis_mat_split_strict_impl(ctx, W, &Lp, &Up, nullptr, nullptr, false);
    if (st != BIS_OK) return cleanup(st);
    const std::vector<int64_t> *level_ptr = nullptr;
    const int32_t *perm = nullptr;
    st = bis_trsv_level_sets(ctx, Lp, &level_ptr, &perm);
    if (st == BIS_OK) {
        const int nl = (int)level_ptr->size() - 1;
        for (int l = 0; l < nl; ++l) {
            const int64_t lo = (*level_ptr)[l], hi = (*level_ptr)[l + 1];
            hipLaunchKernelGGL(ilu0_level_kernel<RP>, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0,
                               ctx->stream, rp, W->col, W->val, dpos, ustart, perm, lo, hi, pivot_tol,
                               pivot_repl, U_D, L_D);
        }
        if (hipGetLastError() != hipSuccess) { ctx->err = "bis_mat_ilu0: level launch failed"; st = BIS_ERR_HIP; }
    }
    if (st == BIS_OK) st = bis_mat_split_strict_impl(ctx, W, Ls_out, Us_out, nullptr, nullptr, false);
    bis_mat_destroy(ctx, Lp);
    bis_mat_destroy(ctx, Up);
    return cleanup(st);
}

} // namespace

extern "C" {

bis_status bis_mat_ilu0(bis_ctx *ctx, const bis_mat *A, double pivot_tol, double pivot_repl,
                        bis_mat **L_strict, bis_mat **U_strict, double *L_D, double *U_D) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A && L_strict && U_strict && L_D && U_D, "bis_mat_ilu0: bad arguments");
    BIS_REQUIRE(ctx, A->n_rows == A->n_cols, "bis_mat_ilu0: square matrix required");
    return A->rp64 ? ilu0_t<int64_t>(ctx, A, pivot_tol, pivot_repl, L_strict, U_strict, L_D, U_D)
                   : ilu0_t<int32_t>(ctx, A, pivot_tol, pivot_repl, L_strict, U_strict, L_D, U_D);
}

} // extern "C"
