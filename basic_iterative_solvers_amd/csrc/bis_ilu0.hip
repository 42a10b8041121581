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
// level; rows of a level are independent.  One WAVE per row (row in LDS, the
// updates of one elimination spread over the lanes; a lane-per-row kernel
// remains for rows beyond the LDS budget).  Every entry sees the serial
// algorithm's fma sequence, so the factors do not depend on the schedule.
// Config-5 stand-in (1.54 M rows, 1.04e8 non-zeros, 1661 levels): 4.45 s with a
// lane per row (35 x 35 x 7 dependent loads per row) -> see DESIGN.md.
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
    bool ascending = true;
    for (int64_t p = s + 1; p < e; ++p) ascending &= col[p - 1] <= col[p];
    if (ascending) { // the usual case: copy
        for (int64_t p = s; p < e; ++p) { wcol[p] = col[p]; wval[p] = val[p]; }
    } else
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

// The same sort with one wave per row: every lane ranks its entries against the whole row
// through cross-lane reads (rows reordered by a permutation are not ascending; the
// lane-per-row rank sort above then costs len^2 dependent loads: 0.1 s on the config-5 stand-in).
template <typename RP>
__global__ __launch_bounds__(256) void sort_rows_wave_kernel(const RP *__restrict__ rp,
                                                             const int32_t *__restrict__ col,
                                                             const double *__restrict__ val, int64_t n,
                                                             int32_t *__restrict__ wcol,
                                                             double *__restrict__ wval,
                                                             int64_t *__restrict__ dpos,
                                                             int64_t *__restrict__ ustart) {
    const int lane = threadIdx.x & 63;
    // grid-stride over the rows: a grid of n/4 workgroups exceeds HIP's 2^32 threads per launch
    // for n > 67 M rows (HPCG-512)
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
    const int64_t s = rp[i], e = rp[i + 1];
    int less_tot = 0, leq_tot = 0, has_diag = 0;
    for (int64_t a0 = s; a0 < e; a0 += 64) {
        const int64_t pa = a0 + lane;
        const bool act = pa < e;
        const int c = act ? col[pa] : INT32_MAX;
        const double v = act ? val[pa] : 0.0;
        int rank = 0;
        for (int64_t b0 = s; b0 < e; b0 += 64) {
            const int64_t pb = b0 + lane;
            const int cb = pb < e ? col[pb] : INT32_MAX;
            const int cnt = (int)(e - b0 < 64 ? e - b0 : 64);
            for (int j = 0; j < cnt; ++j) {
                const int o = __shfl(cb, j, 64);
                rank += (o < c) || (o == c && b0 + j < pa);
            }
        }
        if (act) { wcol[s + rank] = c; wval[s + rank] = v; }
        less_tot += __popcll(__ballot(act && c < i));
        leq_tot += __popcll(__ballot(act && c <= i));
        has_diag |= __ballot(act && c == i) != 0ull;
    }
    if (lane == 0) {
        dpos[i] = has_diag ? s + less_tot : -1; // first entry equal to i in the sorted row
        ustart[i] = s + leq_tot;
    }
    }
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

// Wave-per-row form of the same elimination (rows of up to kIluMaxRow entries):
// the row lives in LDS; the eliminations k stay sequential and ascending (the
// reference's order), but the updates of one elimination -- one per entry of U's
// row k -- run on the lanes of the wave, each lane locating its column in the LDS
// copy by binary search.  Every entry receives the same fma sequence as in the
// serial algorithm, so the factors are identical to the lane-per-row kernel's.
constexpr int kIluMaxRow = 1024;

template <typename RP>
__global__ __launch_bounds__(256) void ilu0_level_wave_kernel(const RP *__restrict__ rp,
                                                              const int32_t *__restrict__ wcol, double *wval,
                                                              const int64_t *__restrict__ dpos,
                                                              const int64_t *__restrict__ ustart,
                                                              const int32_t *__restrict__ perm, int64_t begin,
                                                              int64_t end, double pivot_tol, double pivot_repl,
                                                              double *U_D, double *L_D, int max_row) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ilu_smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *lval = reinterpret_cast<double *>(ilu_smem) + (size_t)wave * max_row;
    int32_t *lcol = reinterpret_cast<int32_t *>(reinterpret_cast<double *>(ilu_smem) + (size_t)4 * max_row) +
                    (size_t)wave * max_row;
    const int64_t t = begin + (int64_t)blockIdx.x * 4 + wave;
    if (t >= end) return; // whole wave
    const int i = perm[t];
    const int64_t s = rp[i], e = rp[i + 1];
    const int len = (int)(e - s);
    for (int q = lane; q < len; q += 64) { lcol[q] = wcol[s + q]; lval[q] = wval[s + q]; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int p = 0; p < len; ++p) {
        const int k = lcol[p]; // uniform
        if (k >= i) break;
        const double pivot = U_D[k]; // row k finished in an earlier level
        if (fabs(pivot) < 1e-16) continue;
        const double factor = lval[p] / pivot; // every lane computes the same value
        const int64_t us = ustart[k], ue = (int64_t)rp[k + 1];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // all lanes have read lval[p] before it is overwritten
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) lval[p] = factor;
        for (int64_t q = us + lane; q < ue; q += 64) {
            const int j = wcol[q];
            const double u = wval[q];
            int lo = p + 1, hi = len; // j > k: search the rest of row i
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (lcol[mid] < j) lo = mid + 1; else hi = mid;
            }
            if (lo < len && lcol[lo] == j) {
                const double w = lval[lo];
                if (w != 0.0) lval[lo] = fma(-factor, u, w);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // updates visible to the next elimination
        __builtin_amdgcn_wave_barrier();
    }
    const int64_t dp = dpos[i];
    double u_diag = dp >= 0 ? lval[dp - s] : 0.0;
    if (fabs(u_diag) < pivot_tol) u_diag = (u_diag >= 0 ? 1.0 : -1.0) * pivot_repl;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0 && dp >= 0) lval[dp - s] = u_diag;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int q = lane; q < len; q += 64) wval[s + q] = lval[q];
    if (lane == 0) { U_D[i] = u_diag; L_D[i] = 1.0; }
}

// The same elimination as ONE launch (persistent grid).  With a launch per level a row cannot begin before the whole previous level
// has ended, although all but its last few eliminations need rows that were finished long ago: a level then costs a row's full time
// (~35 eliminations of two dependent trips to memory each: 48 us per level measured on the config-5 stand-in, 1674 levels = 80 ms).
// Here the rows are dealt round robin, in level order, to the resident waves (static, like sptrsv_wave_kernel: a wave walks its
// positions in ascending order and waits only for rows at smaller positions, so every position completes provided the whole grid
// is resident); a wave waits PER ELIMINATION for the flag of the row it needs -- set by that row's wave after its stores have
// drained -- so only the eliminations that really are on the dependency chain wait.  Hand-off form: the finished row's values and
// pivot are written with sc1 (write-through) stores, `s_waitcnt vmcnt(0)`, then the sc1 flag; the reader polls the flag with sc1
// loads and reads the row with sc1 loads (MI355X_MICROARCH.md, hand-offs measured with sc1 loads in place of the acquire).  The
// arithmetic is the level kernel's, entry for entry.
constexpr unsigned kIluSpin = 1u << 21;
template <typename RP>
__global__ __launch_bounds__(256, 4) void ilu0_persistent_kernel(const RP *__restrict__ rp, const int32_t *__restrict__ wcol, double *wval,
                                                                 const int64_t *__restrict__ dpos, const int64_t *__restrict__ ustart,
                                                                 const int32_t *__restrict__ perm, int64_t n, double pivot_tol,
                                                                 double pivot_repl, double *U_D, double *L_D, int max_row, int *flag,
                                                                 unsigned *fault) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ilu_smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *lval = reinterpret_cast<double *>(ilu_smem) + (size_t)wave * max_row;
    int32_t *lcol = reinterpret_cast<int32_t *>(reinterpret_cast<double *>(ilu_smem) + (size_t)4 * max_row) + (size_t)wave * max_row;
    unsigned long long *wv = reinterpret_cast<unsigned long long *>(wval);
    unsigned long long *ud = reinterpret_cast<unsigned long long *>(U_D);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    bool aborted = false;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n; t += n_waves) {
        const int i = perm[t];
        const int64_t s = rp[i], e = rp[i + 1];
        const int len = (int)(e - s);
        for (int q = lane; q < len; q += 64) { lcol[q] = wcol[s + q]; lval[q] = wval[s + q]; } // (the row's own entries: nobody else writes them)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // An elimination is two dependent trips to memory -- (flag, where U's row k lies), then (pivot, the row's entries) -- and a row has
        // ~35 of them.  What does not depend on row k being finished (its flag's current value, ustart[k], rp[k + 1]) is fetched for the NEXT
        // elimination while this one runs; the pivot and the entries are loaded together once the flag has been seen (never before: loads of
        // one wave may be served out of order, and a pivot read ahead of its flag could be the unfinished one).
        int kn = len > 0 ? lcol[0] : i;
        int fn = 1;
        int64_t usn = 0, uen = 0;
        if (kn < i) {
            fn = __hip_atomic_load(&flag[kn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            usn = ustart[kn];
            uen = (int64_t)rp[kn + 1];
        }
        for (int p = 0; p < len; ++p) {
            const int k = kn; // uniform
            if (k >= i) break;
            int f = fn;
            const int64_t us = usn, ue = uen;
            kn = p + 1 < len ? lcol[p + 1] : i;
            if (kn < i) {
                fn = __hip_atomic_load(&flag[kn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                usn = ustart[kn];
                uen = (int64_t)rp[kn + 1];
            }
            // row k must be finished: its wave set the flag after its stores had drained
            if (!aborted && f == 0) {
                unsigned spins = 0;
                while ((f = __hip_atomic_load(&flag[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                    if (++spins > kIluSpin || ((spins & 1023u) == 0u && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)) {
                        aborted = true; // bounded: the factors are wrong from here on, the context's fault word says so
                        if (lane == 0) __hip_atomic_fetch_or(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            // pivot and the first 64 entries of U's row k in one trip
            const int64_t q0 = us + lane;
            const bool act0 = q0 < ue;
            const double pivot = __longlong_as_double((long long)__hip_atomic_load(&ud[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const int j0 = wcol[act0 ? q0 : us];
            const double u0 = __longlong_as_double((long long)__hip_atomic_load(&wv[act0 ? q0 : us], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (fabs(pivot) < 1e-16) continue;
            const double factor = lval[p] / pivot; // every lane computes the same value
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // all lanes have read lval[p] before it is overwritten
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) lval[p] = factor;
            for (int64_t q = q0; q < ue; q += 64) {
                const int j = q == q0 ? j0 : wcol[q];
                const double u = q == q0 ? u0 : __longlong_as_double((long long)__hip_atomic_load(&wv[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                int lo = p + 1, hi = len; // j > k: search the rest of row i
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (lcol[mid] < j) lo = mid + 1; else hi = mid;
                }
                if (lo < len && lcol[lo] == j) {
                    const double w = lval[lo];
                    if (w != 0.0) lval[lo] = fma(-factor, u, w);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // updates visible to the next elimination
            __builtin_amdgcn_wave_barrier();
        }
        const int64_t dp = dpos[i];
        double u_diag = dp >= 0 ? lval[dp - s] : 0.0;
        if (fabs(u_diag) < pivot_tol) u_diag = (u_diag >= 0 ? 1.0 : -1.0) * pivot_repl;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0 && dp >= 0) lval[dp - s] = u_diag;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (int q = lane; q < len; q += 64)
            __hip_atomic_store(&wv[s + q], (unsigned long long)__double_as_longlong(lval[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
            __hip_atomic_store(&ud[i], (unsigned long long)__double_as_longlong(u_diag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            L_D[i] = 1.0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every lane's stores of this row have left before the flag does
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(&flag[i], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// rows of the sorted copy with a repeated column: the wave-per-row kernel locates ONE entry per column by binary search,
// the reference's serial loop updates every copy -- such matrices take the lane-per-row kernel
template <typename RP>
__global__ __launch_bounds__(256) void dup_check_kernel(const RP *__restrict__ rp, const int32_t *__restrict__ wcol, int64_t n, int *flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool dup = false;
    for (int64_t p = (int64_t)rp[i] + 1; p < (int64_t)rp[i + 1]; ++p) dup |= wcol[p] == wcol[p - 1];
    if (dup) atomicOr(flag, 1);
}

template <typename RP>
bis_status ilu0_t(bis_ctx *ctx, const bis_mat *A, double pivot_tol, double pivot_repl, bis_mat **Ls_out,
                  bis_mat **Us_out, double *L_D, double *U_D) {
    const int64_t n = A->n_rows;
    bis_mat *W = nullptr;
    bis_status st = bis_mat_alloc(ctx, n, A->n_cols, A->nnz, A->rp64, &W);
    if (st != BIS_OK) return st;
    for (int i = 0; i < 4; ++i) W->grid[i] = A->grid[i]; // the factors inherit it through the split
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
    if (n > 0) {
        if (bis_opts().ilu0_wave != 0)
            hipLaunchKernelGGL(sort_rows_wave_kernel<RP>, dim3((unsigned)std::min<int64_t>((n + 3) / 4, 1 << 22)), dim3(256), 0, ctx->stream,
                               (const RP *)A->row_ptr, A->col, A->val, n, W->col, W->val, dpos, ustart);
        else
            hipLaunchKernelGGL(sort_rows_kernel<RP>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const RP *)A->row_ptr, A->col, A->val, n, W->col, W->val, dpos, ustart);
    }
    int has_dups = 0;
    if (n > 0) {
        int *flag = (int *)ctx->counters + 38;
        hipMemsetAsync(flag, 0, sizeof(int), ctx->stream);
        hipLaunchKernelGGL(dup_check_kernel<RP>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, rp, W->col, n, flag);
        e = hipMemcpyAsync(&has_dups, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ctx->err = std::string("bis_mat_ilu0: ") + hipGetErrorString(e); return cleanup(BIS_ERR_HIP); }
    }
    st = bis_mat_finalize(ctx, W);
    if (st != BIS_OK) return cleanup(st);
    // dependency levels = levels of the strict lower triangle of the pattern
    bis_mat *Lp = nullptr, *Up = nullptr;
    st = bis_mat_split_strict_impl(ctx, W, &Lp, &Up, nullptr, nullptr, false);
    if (st != BIS_OK) return cleanup(st);
    const std::vector<int64_t> *level_ptr = nullptr;
    const int32_t *perm = nullptr;
    st = bis_trsv_level_sets(ctx, Lp, &level_ptr, &perm);
    const bool wave_ok = W->max_row_nnz <= kIluMaxRow && bis_opts().ilu0_wave != 0 && !has_dups;
    bool done = false;
    if (st == BIS_OK && wave_ok && bis_opts().ilu0_persistent != 0 && n > 0) {
        // one launch: rows in level order over a resident grid, a flag per finished row (see the kernel)
        const int mr = std::max(W->max_row_nnz, 1);
        const size_t smem = (size_t)4 * mr * (sizeof(double) + sizeof(int32_t));
        int nb = 0;
        hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ilu0_persistent_kernel<RP>, 256, smem);
        (void)hipGetLastError();
        int *flag = nullptr;
        if (oe == hipSuccess && nb > 0 && hipMalloc(&flag, sizeof(int) * (size_t)n) == hipSuccess) {
            const int share = std::max(1, bis_opts().device_share);
            // workgroups per CU: 6 of the 8 the runtime reports resident (round 5; whole call at 2 / 4 / 6 / 7 per CU: `unstr:80,80,80` as
            // generated 72 / 50 / 42 / 40 ms, RCM-ordered 92 / 78 / 74 / 74, fem:80,80,81 79 / 63 / 57 / 56 -- `tools/ilu_ab.py`).  NOT all 8: the
            // rows are dealt statically, every workgroup must be resident, and a grid that needs every slot of every CU does not always
            // get them (8 per CU: the run ended with BIS_ERR_SYNC -- loudly, as designed -- on the first try); option ilu0_wgs
            const int per_cu = std::min(nb, bis_opts().ilu0_wgs > 0 ? bis_opts().ilu0_wgs : 6);
            const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, (int64_t)ctx->n_cus * per_cu / share));
            hipMemsetAsync(flag, 0, sizeof(int) * (size_t)n, ctx->stream);
            hipLaunchKernelGGL(ilu0_persistent_kernel<RP>, dim3((unsigned)grid), dim3(256), smem, ctx->stream, rp, W->col, W->val, dpos, ustart,
                               perm, n, pivot_tol, pivot_repl, U_D, L_D, mr, flag, ctx->fault_dev);
            hipError_t le = hipGetLastError();
            if (le == hipSuccess) le = hipStreamSynchronize(ctx->stream);
            hipFree(flag);
            if (le != hipSuccess) { ctx->err = std::string("bis_mat_ilu0: ") + hipGetErrorString(le); st = BIS_ERR_HIP; }
            else if (bis_status fs = bis_fault_check(ctx)) st = fs;
            done = true;
        } else {
            (void)hipGetLastError();
            hipFree(flag);
        }
    }
    if (st == BIS_OK && !done) {
        const int nl = (int)level_ptr->size() - 1;
        for (int l = 0; l < nl; ++l) {
            const int64_t lo = (*level_ptr)[l], hi = (*level_ptr)[l + 1];
            if (wave_ok) {
                const int mr = std::max(W->max_row_nnz, 1);
                hipLaunchKernelGGL(ilu0_level_wave_kernel<RP>, dim3((unsigned)((hi - lo + 3) / 4)), dim3(256),
                                   (size_t)4 * mr * (sizeof(double) + sizeof(int32_t)), ctx->stream, rp, W->col,
                                   W->val, dpos, ustart, perm, lo, hi, pivot_tol, pivot_repl, U_D, L_D, mr);
            } else
            hipLaunchKernelGGL(ilu0_level_kernel<RP>, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0,
                               ctx->stream, rp, W->col, W->val, dpos, ustart, perm, lo, hi, pivot_tol,
                               pivot_repl, U_D, L_D);
        }
        if (hipGetLastError() != hipSuccess) { ctx->err = "bis_mat_ilu0: level launch failed"; st = BIS_ERR_HIP; }
    }
    if (st == BIS_OK) st = bis_mat_split_strict_impl(ctx, W, Ls_out, Us_out, nullptr, nullptr, false);
    if (st == BIS_OK) bis_trsv_plan_adopt(*Ls_out, Lp, false); // same pattern: L's forward sweep starts from the levels found above
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
