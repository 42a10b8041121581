// bis_cg.hip -- the CG iteration of the reference (methods/cg.hpp:6-54, the
// residual sampling of :162-166 and the stopping test of solver.hpp:177-192)
// as a fused, sync-free device schedule.
//
// Same arithmetic per element as the reference's separate passes:
//   tmp   = A p                                   cg.hpp:16
//   alpha = (r,z) / (tmp,p)                       cg.hpp:19-23
//   x     = x + alpha p ;  r = r - alpha tmp      cg.hpp:28-31
//   z     = M^-1 r  (None: copy, Jacobi: r/D)     cg.hpp:37-41, kernels.hpp:357,398
//   beta  = (r_new,z_new) / (r,z)                 cg.hpp:47
//   p     = z + beta p                            cg.hpp:52
//   ||r_new||_2 recorded, stop test               cg.hpp:162-166, solver.hpp:177-192
// but in three streaming passes separated by the two global reductions
// (12*nnz + 92*N bytes per iteration, +16*N with Jacobi):
//   pass A  SpMV with the (tmp,p) partial sums fused into its epilogue
//   pass B  x/r/z update with the (r,z) and (r,r) partial sums fused
//   pass C  p update
// alpha, beta, the residual history and the stop flag live on the device; the
// host only enqueues iterations and reads the status when it wants to.  Once
// the stop test fires, the remaining enqueued passes are no-ops, so the result
// is exactly the state at the reference's stopping iteration.
#include "bis_internal.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>

struct bis_dist;
bis_status bis_dist_spmv_launch(bis_ctx *ctx, bis_dist *d, double *x_ext, double *y, const double *w,
                                int *n_partials);
int bis_dist_total_blocks(const bis_dist *d);
size_t bis_dist_partials_need(const bis_dist *d);
int64_t bis_dist_n_local(const bis_dist *d);
int64_t bis_dist_n_ext(const bis_dist *d);
bis_status bis_dist_allreduce(bis_ctx *ctx, bis_dist *d, double *buf_dev, int count);

struct bis_cg {
    bis_dist *dist = nullptr;    // non-null: row-partitioned operator
    const bis_mat *A = nullptr;
    const double *A_D = nullptr; // nullptr: no preconditioner
    const double *b = nullptr;
    double *x = nullptr;
    int64_t n = 0;
    double *p = nullptr, *r = nullptr, *z = nullptr, *tmp = nullptr;
    double *sc = nullptr;   // device scalars, see enum below
    double *pap_stage = nullptr; // [kPapBlocks] stage-1 sums of the SpMV's fused partials
    int *flags = nullptr;   // device: [0] iters, [1] done, [2] converged, [3] iteration at which the stop test fired
    double *hist = nullptr; // device residual history
    int hist_cap = 0;
    int enqueued = 0;
    bool initialised = false; // bis_cg_init has run: p0 and (r,z) were made with the preconditioner set at that time
    unsigned *counters = nullptr; // device: arrival tickets of the last-arriver reductions ([0] pap, [1] pass B, [4...] pass B's per-group counters)
    // general preconditioner (bis_cg_set_preconditioner): z = M^-1 r through bis_apply_preconditioner
    int pc = -1;
    const bis_mat *pcL = nullptr, *pcU = nullptr;
    const double *pcAD = nullptr, *pcADinv = nullptr, *pcLD = nullptr, *pcUD = nullptr;
    double *pc_work = nullptr;
    int pc_outer = 1, pc_inner = 0;
};

namespace {

enum { S_RZ = 0, S_PAP, S_RZ_NEW, S_RR, S_ALPHA, S_BETA, S_STOP, S_COUNT = 8 }; // S_RZ_NEW,S_RR adjacent
constexpr int kT = 256;
constexpr int kMaxIters = 1 << 20;

// ---- last-arriver reductions ---------------------------------------------------------
// A streaming pass ends with a global reduction of per-workgroup partial sums.  Instead of a
// second launch, every workgroup publishes its partials (write-through `sc1` stores, drained with
// vmcnt(0) before the arrival is counted -- the "drained sc1 payload, then the flag" hand-off of
// MI355X_MICROARCH.md, no release fence that would write back the pass' own dirty lines) and
// takes a ticket; the workgroup that takes the LAST ticket re-reads all partials with L2-bypassing
// loads and sums them in index order, so the result does not depend on which workgroup finishes
// last: bit-reproducible like the two-launch form.
__device__ __forceinline__ void publish(double *slot, double v) {
    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double fetch(const double *slot) {
    return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// thread 0 only; returns true in the workgroup that arrives last (and re-arms the counter)
__device__ __forceinline__ bool arrive_last(unsigned *counter, unsigned n_groups) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the published partials have left this CU
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t != n_groups - 1) return false;
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // next use: after a kernel boundary
    return true;
}

// The same with a counter per 16 workgroups under one top counter.  Tickets on ONE address serialise at ~11.4 ns each (measured:
// the sweeps' ticket counter): 2048 workgroups of pass B cost 23 us in tickets alone -- nothing next to 70 us of streaming
// on 16.8 M rows, but most of the pass on one rank's 1/8 share of a strong-scaled problem (34 us measured for 2 M rows against
// 10 us of bandwidth time: profiles/r04_b_dist_gap_slab32.txt).  Two levels: 16 + n/16 tickets on the critical path.
constexpr unsigned kArriveGroup = 16;
constexpr unsigned kArriveSubs = 2048 / kArriveGroup; // kMaxReduceBlocks workgroups at most
__device__ __forceinline__ bool arrive_last2(unsigned *top, unsigned *sub, unsigned block, unsigned n_blocks) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the published partials have left this CU
    const unsigned g = block / kArriveGroup;
    const unsigned n_in = min(kArriveGroup, n_blocks - g * kArriveGroup), n_groups = (n_blocks + kArriveGroup - 1) / kArriveGroup;
    if (__hip_atomic_fetch_add(&sub[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != n_in - 1) return false;
    __hip_atomic_store(&sub[g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != n_groups - 1) return false;
    __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // next use: after a kernel boundary
    return true;
}

// the scalar bookkeeping of one iteration (thread 0 of one workgroup): beta, the recorded
// residual norm, the iteration count and the stopping test of solver.hpp:177-192
__device__ __forceinline__ void cg_book(double rz_new, double rr, double *sc, int *flags, double *hist, int hist_cap) {
    const double rz_old = sc[S_RZ];
    sc[S_BETA] = rz_new / rz_old;          // cg.hpp:47
    sc[S_RZ] = rz_new;
    sc[S_RR] = rr;
    const double norm = sqrt(rr);          // cg.hpp:164
    const int it = flags[0] + 1;           // solver_harness.hpp:21
    flags[0] = it;
    if (it < hist_cap) hist[it] = norm;    // solver.hpp:161-163
    // check_stopping_criteria, solver.hpp:177-192 (max_iters is the host's)
    const bool conv = fabs(norm) < sc[S_STOP];
    const bool diverged = fabs(norm) > DBL_MAX || norm != norm;
    if (conv || diverged) { flags[1] = 1; flags[2] = conv ? 1 : 0; flags[3] = it; } // flags[3]: pass C of iteration `it` still updates x
}

// sc[S_PAP] = sum of the SpMV's fused per-wave partials (~1 M of them on HPCG-256): kPapBlocks
// workgroups sum contiguous chunks, the last arriver sums the kPapBlocks chunk sums in order.
constexpr int kPapBlocks = 256;
__global__ __launch_bounds__(256) void cg_finish_pap_kernel(const double *__restrict__ partials, int n_partials,
                                                            double *stage, double *sc, const int *flags,
                                                            unsigned *counter) {
    __shared__ double lds[4];
    __shared__ bool last;
    if (flags[1]) return;
    const int per = (n_partials + kPapBlocks - 1) / kPapBlocks;
    const int lo = blockIdx.x * per, hi = min(lo + per, n_partials);
    double acc = 0.0;
    for (int i = lo + (int)threadIdx.x; i < hi; i += 256) acc += partials[i];
    const double s = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) {
        publish(stage + blockIdx.x, s);
        last = arrive_last(counter, kPapBlocks);
    }
    __syncthreads();
    if (!last) return;
    const double t = block_sum<256>(threadIdx.x < kPapBlocks ? fetch(stage + threadIdx.x) : 0.0, lds);
    if (threadIdx.x == 0) sc[S_PAP] = t;
}

// pass B: r -= alpha tmp; z = r/D (or r); (r,z), (r,r); the last workgroup reduces them and, on one
// GPU, does the iteration's bookkeeping (DIST: the sums go through an all-reduce first, then cg_book_kernel).
typedef double cg_v2d_t __attribute__((ext_vector_type(2)));
template <bool JACOBI, bool DIST>
__global__ __launch_bounds__(kT) void cg_update_kernel(int64_t n, double *sc, int *flags,
                                                       const double *__restrict__ tmp,
                                                       const double *__restrict__ D,
                                                       double *__restrict__ r,
                                                       double *__restrict__ z,
                                                       double *partials, size_t stride, unsigned *counter,
                                                       double *hist, int hist_cap, int nt_tmp, int nt_r) {
    __shared__ double lds[kT / 64];
    __shared__ bool last;
    if (flags[1]) return;
    const double alpha = sc[S_RZ] / sc[S_PAP]; // cg.hpp:19-23
    const int64_t n2 = n >> 1, gs = (int64_t)gridDim.x * kT;
    double rz = 0.0, rr = 0.0;
    const double2 *t2 = reinterpret_cast<const double2 *>(tmp);
    const double2 *D2 = reinterpret_cast<const double2 *>(D);
    double2 *r2 = reinterpret_cast<double2 *>(r);
    double2 *z2 = reinterpret_cast<double2 *>(z);
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n2; i += gs) {
        double2 tv;
        if (nt_tmp) { // A p is dead after this pass (the next SpMV overwrites it): no need to keep its lines
            const cg_v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const cg_v2d_t *>(t2 + i));
            tv = make_double2(v.x, v.y);
        } else tv = t2[i];
        double2 rv, zv;
        if (nt_r) { // (option cg_nt_x = 3: every input stream of the pass read non-temporally, as the BLAS-1 kernels do)
            const cg_v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const cg_v2d_t *>(r2 + i));
            rv = make_double2(v.x, v.y);
        } else rv = r2[i];
        rv.x = fma(-alpha, tv.x, rv.x);        // cg.hpp:31
        rv.y = fma(-alpha, tv.y, rv.y);
        r2[i] = rv;
        if (JACOBI) {
            const double2 dv = D2[i];
            zv.x = rv.x / (1.0 * dv.x);        // kernels.hpp:151
            zv.y = rv.y / (1.0 * dv.y);
            z2[i] = zv;
        } else {
            zv = rv;
        }
        rz = fma(rv.x, zv.x, rz);
        rz = fma(rv.y, zv.y, rz);
        rr = fma(rv.x, rv.x, rr);
        rr = fma(rv.y, rv.y, rr);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double rv = fma(-alpha, tmp[i], r[i]);
        r[i] = rv;
        double zv = rv;
        if (JACOBI) { zv = rv / (1.0 * D[i]); z[i] = zv; }
        rz = fma(rv, zv, rz);
        rr = fma(rv, rv, rr);
    }
    const double s0 = block_sum<kT>(rz, lds);
    __syncthreads();
    const double s1 = block_sum<kT>(rr, lds);
    if (threadIdx.x == 0) {
        publish(partials + blockIdx.x, s0);
        publish(partials + stride + blockIdx.x, s1);
        last = arrive_last2(counter, counter + 3, blockIdx.x, gridDim.x);
    }
    __syncthreads();
    if (!last) return;
    // every other workgroup has read sc[S_RZ] / sc[S_PAP] before it arrived: the scalars may change now
    double a0 = 0.0, a1 = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += kT) { a0 += fetch(partials + i); a1 += fetch(partials + stride + i); }
    const double rz_new = block_sum<kT>(a0, lds);
    __syncthreads();
    const double rr_new = block_sum<kT>(a1, lds);
    if (threadIdx.x == 0) {
        sc[S_ALPHA] = alpha;                   // pass C applies it to x
        if (DIST) { sc[S_RZ_NEW] = rz_new; sc[S_RR] = rr_new; }
        else cg_book(rz_new, rr_new, sc, flags, hist, hist_cap);
    }
}

// distributed schedule: the bookkeeping after the all-reduce of {(r,z), (r,r)}
__global__ void cg_book_kernel(double *sc, int *flags, double *hist, int hist_cap) {
    if (flags[1]) return;
    if (threadIdx.x == 0) cg_book(sc[S_RZ_NEW], sc[S_RR], sc, flags, hist, hist_cap);
}

// pass C: x += alpha p (cg.hpp:28, deferred to here: p is read once for both updates, 8 N bytes
// less per iteration than updating x in pass B); p = z + beta p (cg.hpp:52; None: z == r).
// `it` = the iteration this launch belongs to: when the stop test fired in THIS iteration the x
// update still runs (the reference updates x before it samples the residual), later launches are no-ops.
template <bool NT, bool NT_ALL = false>
__global__ __launch_bounds__(kT) void cg_p_update_kernel(int64_t n, const double *__restrict__ sc,
                                                         const int *__restrict__ flags, int it,
                                                         const double *__restrict__ z,
                                                         double *__restrict__ x,
                                                         double *__restrict__ p) {
    if (flags[1] && flags[3] != it) return;
    const double alpha = sc[S_ALPHA], beta = sc[S_BETA];
    const int64_t n2 = n >> 1, gs = (int64_t)gridDim.x * kT;
    const double2 *z2 = reinterpret_cast<const double2 *>(z);
    double2 *p2 = reinterpret_cast<double2 *>(p);
    double2 *x2 = reinterpret_cast<double2 *>(x);
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n2; i += gs) {
        double2 zv, pv, xv;
        if (NT_ALL) { // (option cg_nt_x = 3)
            const cg_v2d_t a = __builtin_nontemporal_load(reinterpret_cast<const cg_v2d_t *>(z2 + i));
            const cg_v2d_t c = __builtin_nontemporal_load(reinterpret_cast<const cg_v2d_t *>(p2 + i));
            zv = make_double2(a.x, a.y); pv = make_double2(c.x, c.y);
        } else { zv = z2[i]; pv = p2[i]; }
        if (NT) { // x is touched here and nowhere else in the iteration: keep it out of the caches p, r and A p live in
            const cg_v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const cg_v2d_t *>(x2 + i));
            xv = make_double2(v.x, v.y);
        } else xv = x2[i];
        xv.x = fma(alpha, pv.x, xv.x);
        xv.y = fma(alpha, pv.y, xv.y);
        pv.x = fma(beta, pv.x, zv.x);
        pv.y = fma(beta, pv.y, zv.y);
        if (NT) { cg_v2d_t v; v.x = xv.x; v.y = xv.y; __builtin_nontemporal_store(v, reinterpret_cast<cg_v2d_t *>(x2 + i)); }
        else x2[i] = xv;
        p2[i] = pv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        x[n - 1] = fma(alpha, p[n - 1], x[n - 1]);
        p[n - 1] = fma(beta, p[n - 1], z[n - 1]);
    }
}

inline int grid_for(int64_t n) {
    int64_t g = ((n >> 1) + kT - 1) / kT;
    if (g < 1) g = 1;
    if (g > kMaxReduceBlocks) g = kMaxReduceBlocks;
    return (int)g;
}

} // namespace

extern "C" {

static bis_status cg_create_common(bis_ctx *ctx, bis_dist *dist, const bis_mat *A, const double *A_D,
                                   const double *b, double *x, bis_cg **out) {
    bis_cg *cg = new bis_cg;
    cg->dist = dist;
    cg->A = A;
    cg->A_D = A_D;
    cg->b = b;
    cg->x = x;
    cg->n = dist ? bis_dist_n_local(dist) : A->n_rows;
    const int64_t n_ext = dist ? bis_dist_n_ext(dist) : cg->n; // p is the SpMV input: needs the halo tail
    cg->hist_cap = 1 << 16;
    bis_status st = bis_vec_alloc(ctx, n_ext, &cg->p);
    if (st == BIS_OK) st = bis_vec_alloc(ctx, cg->n, &cg->r);
    if (st == BIS_OK && A_D) st = bis_vec_alloc(ctx, cg->n, &cg->z);
    if (st == BIS_OK) st = bis_vec_alloc(ctx, cg->n, &cg->tmp);
    if (st == BIS_OK) st = bis_vec_alloc(ctx, S_COUNT, &cg->sc);
    if (st == BIS_OK) st = bis_vec_alloc(ctx, kPapBlocks, &cg->pap_stage);
    if (st == BIS_OK) st = bis_vec_alloc(ctx, cg->hist_cap, &cg->hist);
    if (st == BIS_OK && hipMalloc(&cg->flags, sizeof(int) * 4) != hipSuccess) st = BIS_ERR_HIP;
    if (st == BIS_OK && (hipMalloc(&cg->counters, sizeof(unsigned) * (4 + kArriveSubs)) != hipSuccess ||
                         hipMemsetAsync(cg->counters, 0, sizeof(unsigned) * (4 + kArriveSubs), ctx->stream) != hipSuccess)) st = BIS_ERR_HIP;
    if (st != BIS_OK) { bis_cg_destroy(ctx, cg); return st; }
    if (!A_D) cg->z = cg->r; // z aliases r without a preconditioner
    *out = cg;
    return BIS_OK;
}

bis_status bis_cg_create(bis_ctx *ctx, const bis_mat *A, const double *A_D, const double *b,
                         double *x, bis_cg **out) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A && b && x && out, "bis_cg_create: bad arguments");
    BIS_REQUIRE(ctx, A->n_rows == A->n_cols, "bis_cg_create: square matrix required");
    return cg_create_common(ctx, nullptr, A, A_D, b, x, out);
}

bis_status bis_dist_cg_create(bis_ctx *ctx, bis_dist *d, const double *A_D, const double *b,
                              double *x, bis_cg **out) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, d && b && x && out, "bis_dist_cg_create: bad arguments");
    return cg_create_common(ctx, d, nullptr, A_D, b, x, out);
}

bis_status bis_cg_set_preconditioner(bis_ctx *ctx, bis_cg *cg, int precond_type, const bis_mat *L_strict, const bis_mat *U_strict,
                                     const double *A_D, const double *A_D_inv, const double *L_D, const double *U_D,
                                     int outer_iters, int inner_iters) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, cg && precond_type >= BIS_PC_NONE && precond_type <= BIS_PC_ILU0 && outer_iters >= 1 && inner_iters >= 0,
                "bis_cg_set_preconditioner: bad arguments");
    BIS_REQUIRE(ctx, !cg->initialised && cg->enqueued == 0, "bis_cg_set_preconditioner: call it before bis_cg_init / bis_cg_iterate");
    if (cg->z == cg->r) { // z aliased r (no preconditioner at creation): it needs its own storage now
        cg->z = nullptr;
        bis_status st = bis_vec_alloc(ctx, cg->n, &cg->z);
        if (st != BIS_OK) { cg->z = cg->r; return st; }
    }
    if (!cg->pc_work) { bis_status st = bis_vec_alloc(ctx, cg->n, &cg->pc_work); if (st != BIS_OK) return st; }
    cg->pc = precond_type;
    cg->pcL = L_strict; cg->pcU = U_strict;
    cg->pcAD = A_D; cg->pcADinv = A_D_inv; cg->pcLD = L_D; cg->pcUD = U_D;
    cg->pc_outer = outer_iters; cg->pc_inner = inner_iters;
    return BIS_OK;
}

bis_status bis_cg_destroy(bis_ctx *ctx, bis_cg *cg) {
    BIS_CTX_OK(ctx);
    if (!cg) return BIS_OK;
    hipStreamSynchronize(ctx->stream);
    hipFree(cg->p);
    hipFree(cg->r);
    if (cg->z != cg->r) hipFree(cg->z);
    hipFree(cg->tmp);
    hipFree(cg->sc);
    hipFree(cg->pap_stage);
    hipFree(cg->hist);
    hipFree(cg->flags);
    hipFree(cg->counters);
    hipFree(cg->pc_work);
    delete cg;
    return BIS_OK;
}

bis_status bis_cg_init(bis_ctx *ctx, bis_cg *cg, double tol, double *r0_norm_host) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, cg, "bis_cg_init: null handle");
    const int64_t n = cg->n;
    // init_residual, cg.hpp:100-118
    bis_status st;
    double rz = 0.0, rr = 0.0;
    if (cg->dist) {
        st = bis_copy_vector(ctx, cg->p, cg->x, n); // SpMV input needs the halo tail: stage x in p
        if (st == BIS_OK) st = bis_dist_spmv_launch(ctx, cg->dist, cg->p, cg->tmp, nullptr, nullptr);
        if (st == BIS_OK) st = bis_subtract_vectors(ctx, cg->r, cg->b, cg->tmp, n, 1.0);
    } else {
        st = bis_compute_residual(ctx, cg->A, cg->x, cg->b, cg->r, cg->tmp);
    }
    if (st == BIS_OK && cg->pc >= 0)
        st = bis_apply_preconditioner(ctx, cg->pc, n, cg->pcL, cg->pcU, cg->pcAD, cg->pcADinv, cg->pcLD, cg->pcUD, cg->z, cg->r, cg->tmp,
                                      cg->pc_work, cg->pc_outer, cg->pc_inner);
    else if (st == BIS_OK && cg->A_D) st = bis_elemwise_div_vectors(ctx, cg->z, cg->r, cg->A_D, n, 1.0);
    if (st == BIS_OK) st = bis_copy_vector(ctx, cg->p, cg->z, n);
    if (cg->dist) {
        if (st == BIS_OK) st = bis_dist_dot(ctx, cg->dist, cg->r, cg->z, ctx->scalars_dev + 8, &rz);
        if (st == BIS_OK) st = bis_dist_dot(ctx, cg->dist, cg->r, cg->r, ctx->scalars_dev + 8, &rr);
    } else {
        if (st == BIS_OK) st = bis_dot(ctx, cg->r, cg->z, n, &rz);
        if (st == BIS_OK) st = bis_dot(ctx, cg->r, cg->r, n, &rr);
    }
    if (st != BIS_OK) return st;
    const double norm0 = sqrt(rr);
    double sc[S_COUNT] = {0};
    sc[S_RZ] = rz;
    sc[S_RR] = rr;
    sc[S_STOP] = tol * norm0; // init_stopping_criteria, solver.hpp:173-175
    int flags[4] = {0, 0, 0, 0};
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(cg->sc, sc, sizeof sc, hipMemcpyHostToDevice, ctx->stream));
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(cg->flags, flags, sizeof flags, hipMemcpyHostToDevice, ctx->stream));
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(cg->hist, &norm0, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    cg->enqueued = 0;
    cg->initialised = true;
    if (r0_norm_host) *r0_norm_host = norm0;
    return BIS_OK;
}

// one CG iteration enqueued on ctx->stream: 4 launches on one GPU (SpMV with the fused (Ap,p) partials,
// their two-stage sum, pass B with the reductions and the bookkeeping in its last workgroup, pass C)
static bis_status cg_enqueue_iteration(bis_ctx *ctx, bis_cg *cg, int g, int it) {
    const int64_t n = cg->n;
    int n_part = 0;
    bis_status st;
    if (cg->dist) st = bis_dist_spmv_launch(ctx, cg->dist, cg->p, cg->tmp, cg->p, &n_part);
    else st = bis_spmv_launch(ctx, cg->A, cg->p, cg->tmp, cg->p, &n_part);
    if (st != BIS_OK) return st;
    hipLaunchKernelGGL(cg_finish_pap_kernel, dim3(kPapBlocks), dim3(256), 0, ctx->stream,
                       ctx->partials, n_part, cg->pap_stage, cg->sc, cg->flags, cg->counters);
    if (cg->dist) {
        st = bis_dist_allreduce(ctx, cg->dist, cg->sc + S_PAP, 1);
        if (st != BIS_OK) return st;
    }
#define BIS_CG_UPDATE(J, D)                                                                                  \
    hipLaunchKernelGGL((cg_update_kernel<J, D>), dim3(g), dim3(kT), 0, ctx->stream, n, cg->sc, cg->flags,    \
                       cg->tmp, cg->A_D, cg->r, cg->z, ctx->partials, (size_t)kMaxReduceBlocks,              \
                       cg->counters + 1, cg->hist, cg->hist_cap, bis_opts().cg_nt_x != 0 && bis_opts().cg_nt_x != 2, bis_opts().cg_nt_x == 3)
    if (cg->pc >= 0) BIS_CG_UPDATE(false, true); // r update and (r,r) only; z and (r,z) follow below
    else if (cg->dist) { if (cg->A_D) BIS_CG_UPDATE(true, true); else BIS_CG_UPDATE(false, true); }
    else { if (cg->A_D) BIS_CG_UPDATE(true, false); else BIS_CG_UPDATE(false, false); }
#undef BIS_CG_UPDATE
    if (cg->pc >= 0) { // general preconditioner: z = M^-1 r (triangular sweeps ...), then (r,z) with the stream-ordered dot
        st = bis_apply_preconditioner(ctx, cg->pc, n, cg->pcL, cg->pcU, cg->pcAD, cg->pcADinv, cg->pcLD, cg->pcUD, cg->z, cg->r, cg->tmp,
                                      cg->pc_work, cg->pc_outer, cg->pc_inner);
        if (st == BIS_OK) st = bis_dot_dev(ctx, cg->r, cg->z, n, cg->sc + S_RZ_NEW);
        if (st != BIS_OK) return st;
        if (!cg->dist)
            hipLaunchKernelGGL(cg_book_kernel, dim3(1), dim3(64), 0, ctx->stream, cg->sc, cg->flags, cg->hist, cg->hist_cap);
    }
    if (cg->dist) {
        st = bis_dist_allreduce(ctx, cg->dist, cg->sc + S_RZ_NEW, 2); // {(r,z),(r,r)} batched
        if (st != BIS_OK) return st;
        hipLaunchKernelGGL(cg_book_kernel, dim3(1), dim3(64), 0, ctx->stream, cg->sc, cg->flags, cg->hist, cg->hist_cap);
    }
    if (bis_opts().cg_nt_x == 3) hipLaunchKernelGGL((cg_p_update_kernel<true, true>), dim3(g), dim3(kT), 0, ctx->stream, n, cg->sc, cg->flags, it, cg->z, cg->x, cg->p);
    else if (bis_opts().cg_nt_x != 0) hipLaunchKernelGGL(cg_p_update_kernel<true>, dim3(g), dim3(kT), 0, ctx->stream, n, cg->sc, cg->flags, it, cg->z, cg->x, cg->p);
    else hipLaunchKernelGGL(cg_p_update_kernel<false>, dim3(g), dim3(kT), 0, ctx->stream, n, cg->sc, cg->flags, it, cg->z, cg->x, cg->p);
    return BIS_OK;
}

bis_status bis_cg_iterate(bis_ctx *ctx, bis_cg *cg, int n_iters) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, cg && n_iters >= 0 && cg->enqueued + n_iters < kMaxIters,
                "bis_cg_iterate: bad arguments");
    const int64_t n = cg->n;
    if (n == 0) return BIS_OK;
    const int g = grid_for(n);
    const size_t nblk = cg->dist ? (size_t)bis_dist_total_blocks(cg->dist) : (size_t)cg->A->n_blocks_f;
    size_t need = std::max((size_t)2 * kMaxReduceBlocks, nblk * 16); // <= 16 waves per row block
    if (!cg->dist) need = std::max(need, (size_t)bis_spmv_sellwin_slices(cg->A)); // sliced-ELL form: one partial per 64 rows
    else need = std::max(need, bis_dist_partials_need(cg->dist));                 // (three row ranges, each in its own format)
    bis_status st = bis_ensure_partials(ctx, need);
    if (st != BIS_OK) return st;
    ctx->spmv_stop = cg->flags;
    struct StopGuard { bis_ctx *c; ~StopGuard() { c->spmv_stop = nullptr; } } stop_guard{ctx};
    // (replaying a captured iteration as a hipGraph was measured and removed: no gain on ROCm 7.2 --
    // HPCG-32/64/128/256: 2.33/4.83/40.6/607 ms of plain launches against 2.66/5.73/42.8/588 ms of replays)
    for (int done = 0; done < n_iters; ++done) {
        st = cg_enqueue_iteration(ctx, cg, g, cg->enqueued + done + 1);
        if (st != BIS_OK) { cg->enqueued += done; return st; } // (pass C's iteration numbers must keep matching the device's count)
    }
    BIS_HIP_CHECK(ctx, hipGetLastError());
    cg->enqueued += n_iters;
    return BIS_OK;
}

bis_status bis_cg_status(bis_ctx *ctx, bis_cg *cg, int *iters, int *converged, double *hist_host,
                         int hist_cap) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, cg, "bis_cg_status: null handle");
    int flags[4] = {0, 0, 0, 0};
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(flags, cg->flags, sizeof flags, hipMemcpyDeviceToHost, ctx->stream));
    BIS_SYNC_CHECK(ctx);
    if (iters) *iters = flags[0];
    if (converged) *converged = flags[2];
    if (hist_host && hist_cap > 0) {
        int cnt = flags[0] + 1;
        if (cnt > hist_cap) cnt = hist_cap;
        if (cnt > cg->hist_cap) cnt = cg->hist_cap;
        BIS_HIP_CHECK(ctx, hipMemcpyAsync(hist_host, cg->hist, sizeof(double) * (size_t)cnt,
                                          hipMemcpyDeviceToHost, ctx->stream));
        BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return BIS_OK;
}

} // extern "C"
