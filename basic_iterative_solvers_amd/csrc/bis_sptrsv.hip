// bis_sptrsv.hip -- sparse triangular solves (reference kernels.hpp:54-117,
// serial there) and the preconditioner dispatcher built on them
// (kernels.hpp:312-414).
//
// Three execution modes, chosen from the structure of the triangle (get_plan):
//
//  (1) colour-sorted matrices (`-perm mc`): at most 64 contiguous blocks of mutually
//      independent rows (bis_analysis.hip finds them on the device) -- one streaming
//      SpMV launch with the triangular epilogue per block (bis_spmv.hip MODE 2).
//  (2) everything else (natural orderings: hundreds to thousands of dependency
//      levels): level-scheduled, synchronisation-free solve in ONE launch.
//      analysis (once per matrix, on the device): level[r] = 1 + max level of the
//        rows r depends on; rows stably sorted by level into perm[] so that every
//        dependency of the row at position i sits at a position < i.
//      solve: the rows are walked in level order by a persistent grid; the lane
//        (rows of up to 8 dependencies, 256-position tickets from a device counter)
//        or the wave (longer rows, positions dealt round-robin) that owns a row
//        accumulates acc = fma(val, x[col], acc) in CRS storage order -- the
//        reference's natural-order arithmetic exactly -- then x[r] = (b[r]-acc)/D[r],
//        and waits for each x[col] it needs.  Readiness travels with the data:
//        results are published into a scratch vector pre-filled with a NaN sentinel,
//        kept in LEVEL order, by ONE 8-byte sc1 store per row, and consumers poll
//        exactly that word with agent-scope relaxed loads (cdna_hip_programming.md
//        Guideline 16, form R2 "the data IS the flag").  Every dependency belongs to
//        an earlier position, whose owner is resident: progress is guaranteed; every
//        spin is bounded (a lost hand-off would publish NaN instead of hanging).
//        The publishing store is predicated inside volatile asm on the straight-line
//        path of the wait loop -- see the hazard notes at the kernels.
//  (3) at most 64 levels that are not contiguous: one plain launch per level.
//
//   The user-visible x is written with a plain store (nobody polls it), so x
//   may alias b (gmres.hpp:173, gauss_seidel.hpp:37).
//
// HBM traffic ~ 12*nnz_T + 28*N algorithmic (+ 24*N for the sentinel scratch);
// mode (2) is latency-bound on stencils: one cross-CU hand-off (~3 us) per level.
#include "bis_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

struct bis_trsv_plan {
    int32_t *perm = nullptr; // device, rows sorted by level
    double *xs = nullptr;    // device scratch, sentinel-filled before each solve
    bool no_pos = false;
    int32_t *pcol = nullptr; // device: position (in perm) of every column -- the scratch is kept in level order
    int *level = nullptr;    // device: level of every row, kept until the chained sweep's plan has been tried (bis_trsv_chain.hip)
    unsigned *ticket = nullptr;
    int n_levels = 0;
    int64_t n = 0;
    int64_t max_level_width = 0;
    std::vector<int64_t> level_ptr; // host: positions of the level boundaries in perm
    // few-level orderings whose levels are contiguous row ranges: one row view per level
    std::vector<bis_mat *> level_views;
    std::vector<int64_t> level_row0;
    int wave_choice = 0;   // level-scheduled kernels: 0 not decided yet, 1 a wave per row, 2 a lane per row (trial at the first sweep, see trsv_solve_impl)
    float trial_ms[2] = {0.f, 0.f};
};

void bis_trsv_plan_destroy(bis_trsv_plan *p) {
    if (!p) return;
    hipFree(p->perm);
    hipFree(p->xs);
    hipFree(p->pcol);
    hipFree(p->level);
    hipFree(p->ticket);
    for (bis_mat *v : p->level_views) { // row views: only their block tables are theirs
        bis_mat_free_meta(v);
        delete v;
    }
    delete p;
}

void bis_trsv_plan_adopt(bis_mat *to, bis_mat *from, bool backward) {
    if (!to || !from || to == from) return;
    bis_trsv_plan *&src = backward ? from->plan_bwd : from->plan_fwd;
    bis_trsv_plan *&dst = backward ? to->plan_bwd : to->plan_fwd;
    if (!src || dst || !src->level_views.empty() || to->n_rows != from->n_rows || to->nnz != from->nnz || to->rp64 != from->rp64) return;
    dst = src;
    src = nullptr;
}

namespace {

constexpr unsigned long long kSentinel = 0x7FF85EA71E55C0DEull; // quiet NaN + payload
constexpr unsigned long long kCanonNaN = 0x7FF8000000000000ull;
constexpr int kTrsvT = 256;
constexpr unsigned kSpinLimit = 1u << 20; // polls of one row before it gives up and publishes NaN (about a second)
constexpr unsigned kFaultPollMask = 1023u; // a waiting row reads the context's fault word every 1024 polls: once ANY wait of the
                                           // sweep has given up, every other wait ends within a millisecond and later rows do not
                                           // wait at all -- a starved or lost hand-off drains the grid at once instead of row by row
__device__ __forceinline__ bool fault_raised(const unsigned *fault) {
    return __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
}
constexpr int kWaveBlocksPerCU = 4;       // wave-per-row kernel: resident workgroups per CU the launch bound guarantees

// ... and the sweep's ticket counter / elected XCD (none yet): set on the device, in stream order by construction
__global__ __launch_bounds__(256) void fill_sentinel_kernel(unsigned long long *xs, int64_t n, unsigned *ticket) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { ticket[0] = 0u; ticket[1] = 0xffffffffu; }
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) xs[i] = kSentinel;
}

// ONE_XCD: the persistent grid elects the XCD of the first workgroup that arrives
// (HW_REG_XCC_ID, hardware truth -- not a blockIdx convention) and every workgroup
// on another XCD leaves at once.  All participants then share ONE L2, so results
// are published with plain stores (the line stays in that L2) and polled with
// L1-bypassing loads served by the same L2: a hand-off costs an L2 round trip
// instead of two trips through the fabric.  Any placement is correct (at least the
// electing workgroup takes part, tickets are only taken by participants); the
// placement decides only how many workgroups help.
template <typename RP, bool ONE_XCD, int kBatch>
__global__ __launch_bounds__(kTrsvT) void sptrsv_syncfree_kernel(
    const RP *__restrict__ row_ptr, const int32_t *__restrict__ dep /* col, or positions (pcol) */,
    const double *__restrict__ val, const int32_t *__restrict__ perm, int64_t n,
    const double *__restrict__ D, const double *b, double *x, unsigned long long *xs,
    unsigned *ticket, int by_pos, unsigned *fault, const int *stop) {
    __shared__ unsigned s_ticket;
    if (stop && stop[1]) return; // the device schedule this sweep belongs to has stopped: x stays as it is
    if (ONE_XCD) {
        if (threadIdx.x == 0) {
            unsigned my;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my));
            my &= 0xfu;
            const unsigned prev = atomicCAS(ticket + 1, 0xffffffffu, my);
            s_ticket = (prev == 0xffffffffu || prev == my) ? 1u : 0u;
        }
        __syncthreads();
        const unsigned go = s_ticket;
        __syncthreads();
        if (!go) return;
    }
    bool aborted = false; // this lane has seen the sweep fail: its later rows do not wait any more
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        const unsigned t = s_ticket;
        __syncthreads();
        const int64_t base = (int64_t)t * kTrsvT;
        if (base >= n) return; // every wave reaches this once tickets run out
        const int64_t pos = base + threadIdx.x;
        // No `if (pos < n)` around the row: a divergent branch that ends at the tail of the ticket
        // loop may be merged with the back edge, and the skipping lanes would then reach the next
        // trip's barrier ahead of their wave.  Lanes past the end run an empty row, stores masked.
        const bool valid = pos < n;
        {
            const int r = valid ? perm[pos] : 0;
            int64_t k = valid ? (int64_t)row_ptr[r] : 0;
            const int64_t e = valid ? (int64_t)row_ptr[r + 1] : 0;
            const double rhs = valid ? b[r] : 0.0, d = valid ? D[r] : 1.0;
            const int64_t slot = by_pos ? pos : (int64_t)r; // where this row's result is published
            double acc = 0.0;
            unsigned spins = aborted ? kSpinLimit : 0u;
            // One loop, bounded work per trip (lanes of one wave may depend on each
            // other, so no lane may spin in a loop of its own): a batch of up to
            // kBatch dependencies is loaded with independent loads (one round trip),
            // the ready prefix is consumed IN ORDER (the sum keeps CRS order), and
            // only the words that were still pending are polled again, together.
            unsigned long long bits[kBatch];
            double av[kBatch];
            int pc[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) { bits[j] = kSentinel; av[j] = 0.0; pc[j] = 0; }
            int ready = 0, in_batch = 0; // ready == in_batch: batch exhausted, load the next one
            // NOTE the publishing store must execute INSIDE the loop: a block that always
            // leaves the loop (a `break`, or a flag the compiler can see through) is an exit
            // block, and exit blocks run only after EVERY lane of the wave has left the
            // loop -- the store would wait for the very lanes that wait for it.
            bool done = false;
            while (!done) {
                bool publish = false;
                unsigned long long out = kCanonNaN;
                if (ready == in_batch) { // batch consumed
                    k += in_batch;
                    if (k == e) {
                        const double v = (rhs - acc) / d;
                        out = (unsigned long long)__double_as_longlong(v);
                        if (v != v) out = kCanonNaN; // never publish the sentinel pattern
                        publish = true;
                        in_batch = ready = 0;
                    } else {
                        in_batch = e - k < (int64_t)kBatch ? (int)(e - k) : kBatch;
#pragma unroll
                        for (int j = 0; j < kBatch; ++j) {
                            pc[j] = j < in_batch ? dep[k + j] : 0;
                            av[j] = j < in_batch ? val[k + j] : 0.0; // does not wait for the dependency
                        }
#pragma unroll
                        for (int j = 0; j < kBatch; ++j)
                            bits[j] = j < in_batch ? __hip_atomic_load(&xs[pc[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                   : 0ull;
                        ready = 0;
                    }
                } else { // poll again exactly the words of this batch that were still pending
#pragma unroll
                    for (int j = 0; j < kBatch; ++j)
                        if (j >= ready && j < in_batch && bits[j] == kSentinel)
                            bits[j] = __hip_atomic_load(&xs[pc[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    if (ready == j && j < in_batch && bits[j] != kSentinel) {
                        acc = fma(av[j], __longlong_as_double((long long)bits[j]), acc);
                        ready = j + 1;
                    }
                }
                if (ready < in_batch) { // still pending
                    ++spins;
                    if ((spins & kFaultPollMask) == 0u && fault_raised(fault)) spins = kSpinLimit + 1u; // somebody gave up: so do we
                    if (spins > kSpinLimit) { // bounded: a lost hand-off must not hang the GPU -- publishes NaN and
                        publish = true;       // raises the context's fault word (the next blocking call fails)
                        aborted = true;
                        __hip_atomic_fetch_or(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    } else __builtin_amdgcn_s_sleep(1);
                }
                {
                    // The publishing store is predicated INSIDE one volatile asm, with no branch
                    // the compiler can see.  Written as `if (publish) store` the optimiser
                    // (legally, for one thread) turns the wait loop into
                    // "do { poll } while (!publish); store": the store then runs only after
                    // EVERY lane of the wave is ready to publish, and lanes of one wave that
                    // depend on each other wait forever (observed: ROCm 7.2, both with a plain
                    // agent-scope atomic store and with an asm store inside the if).
                    unsigned long long *dst = &xs[slot];
                    const unsigned pflag = (publish && valid) ? 1u : 0u;
                    unsigned long long saved_exec;
                    if (ONE_XCD)
                        asm volatile("v_cmp_ne_u32_e32 vcc, 0, %3\n\ts_and_saveexec_b64 %0, vcc\n\t"
                                     "global_store_dwordx2 %1, %2, off\n\ts_mov_b64 exec, %0"
                                     : "=&s"(saved_exec) : "v"(dst), "v"(out), "v"(pflag) : "vcc", "memory");
                    else
                        asm volatile("v_cmp_ne_u32_e32 vcc, 0, %3\n\ts_and_saveexec_b64 %0, vcc\n\t"
                                     "global_store_dwordx2 %1, %2, off sc1\n\ts_mov_b64 exec, %0"
                                     : "=&s"(saved_exec) : "v"(dst), "v"(out), "v"(pflag) : "vcc", "memory");
                }
                if (publish) {
                    if (valid) x[r] = __longlong_as_double((long long)out); // nobody polls x: may sink out of the loop
                    done = true;
                }
            }
        }
    }
}

// Wave-per-row form for rows of more than 8 dependencies: the 64 lanes of a
// wave load and poll up to 64 dependencies of ONE row together -- every pending
// word is refreshed on every trip at O(1) instructions, no batches in sequence --
// and the sum is then accumulated in CRS order by a scalar loop over the lanes
// (v_readlane), i.e. the reference's fma chain exactly.  Waves take tickets (one
// row each) in level order; lanes of a wave never wait for each other, so the
// wait loop is an ordinary loop here.
template <typename RP>
__global__ __launch_bounds__(kTrsvT, kWaveBlocksPerCU) void sptrsv_wave_kernel(
    const RP *__restrict__ row_ptr, const int32_t *__restrict__ dep, const double *__restrict__ val,
    const int32_t *__restrict__ perm, int64_t n, const double *__restrict__ D, const double *b, double *x,
    unsigned long long *xs, unsigned *ticket, int by_pos, unsigned *fault, const int *stop) {
    if (stop && stop[1]) return;
    const int lane = threadIdx.x & 63;
    // Static round robin over the level-sorted positions, no ticket counter (one atomic
    // per row on one address caps the sweep at ~88 M rows/s: 11.4 ns per row measured).
    // Progress: a wave walks its positions in ascending order and waits only for smaller
    // positions, so by induction every position completes PROVIDED all waves of the
    // grid are resident -- the launch keeps the grid at <= 4 workgroups per CU.
    const int64_t n_waves = (int64_t)gridDim.x * (kTrsvT / 64);
    const int64_t wave0 = (int64_t)blockIdx.x * (kTrsvT / 64) + (threadIdx.x >> 6);
    bool aborted = false; // the sweep has failed (this wave or another gave up): no further waiting
    for (int64_t pos = wave0; pos < n; pos += n_waves) {
        const int r = perm[pos];
        const int64_t s = (int64_t)row_ptr[r], e = (int64_t)row_ptr[r + 1];
        const double rhs = b[r], d = D[r];
        double acc = 0.0;
        bool lost = false;
        for (int64_t k0 = s; k0 < e; k0 += 64) {
            const int64_t k = k0 + lane;
            const bool active = k < e;
            const int pc = active ? dep[k] : 0;
            const double av = active ? val[k] : 0.0;
            unsigned long long v = active ? __hip_atomic_load(&xs[pc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            // fold the ready prefix into the sum while the later words are still awaited (CRS order kept:
            // entries are consumed strictly left to right), so that only the tail remains after the last arrival
            const int cnt = (int)(e - k0 < 64 ? e - k0 : 64);
            unsigned spins = 0;
            int folded = 0;
            // The chain acc_j = fma(a_j, x_j, acc_{j-1}) runs ACROSS the lanes: in step j every lane
            // takes its left neighbour's accumulator (DPP wave_shr:1, lane 0 takes the carry of the
            // previous chunk) and applies its own fma; lane j's value is final after step j and
            // recomputing it later reproduces it.  ~20 cycles per step against ~80 for a scalar loop
            // over v_readlane.
            double lacc = 0.0;
            for (;;) {
                const unsigned long long pend = __ballot(active && v == kSentinel);
                const int upto = pend ? (int)__builtin_ctzll(pend) : cnt;
                const double xv = __longlong_as_double((long long)v);
                for (int j = folded; j < upto; ++j) {
                    const unsigned long long cur = (unsigned long long)__double_as_longlong(lacc);
                    const unsigned long long seed = (unsigned long long)__double_as_longlong(acc); // carry into lane 0
                    const unsigned tlo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)seed, (int)(unsigned)cur, 0x138, 0xf, 0xf, false);
                    const unsigned thi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(seed >> 32), (int)(unsigned)(cur >> 32), 0x138, 0xf, 0xf, false);
                    const double left = __longlong_as_double((long long)(((unsigned long long)thi << 32) | tlo));
                    lacc = fma(av, xv, left);
                }
                folded = upto;
                if (!pend) break;
                ++spins;
                if (!aborted && (spins & kFaultPollMask) == 0u) aborted = __builtin_amdgcn_readfirstlane((int)fault_raised(fault)) != 0;
                if (aborted || spins > kSpinLimit) { // bounded: publishes NaN below and raises the context's fault word
                    lost = true;
                    aborted = true;
                    if (lane == 0) __hip_atomic_fetch_or(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                if (active && v == kSentinel) v = __hip_atomic_load(&xs[pc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_sleep(1);
            }
            // the chunk's result sits in its last lane
            {
                const unsigned long long lb = (unsigned long long)__double_as_longlong(lacc);
                const unsigned long long fin = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(lb >> 32), cnt - 1) << 32) |
                                               (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)lb, cnt - 1);
                acc = __longlong_as_double((long long)fin);
            }
        }
        const double res = (rhs - acc) / d;
        unsigned long long out = (unsigned long long)__double_as_longlong(res);
        if (res != res || lost) out = kCanonNaN; // never publish the sentinel pattern
        // Lane 0 publishes -- predicated inside volatile asm, not `if (lane == 0)`: a
        // divergent branch at the tail of the ticket loop lets the compiler send the other
        // 63 lanes into the next trip without lane 0 (they never take a ticket and spin
        // on ticket 0 forever, and lane 0's store waits for them at the loop exit).
        {
            unsigned long long *dst = &xs[by_pos ? pos : (int64_t)r];
            unsigned long long *dx = reinterpret_cast<unsigned long long *>(&x[r]);
            const unsigned pflag = lane == 0 ? 1u : 0u;
            unsigned long long saved_exec;
            asm volatile("v_cmp_ne_u32_e32 vcc, 0, %4\n\ts_and_saveexec_b64 %0, vcc\n\t"
                         "global_store_dwordx2 %1, %3, off sc1\n\tglobal_store_dwordx2 %2, %3, off\n\t"
                         "s_mov_b64 exec, %0"
                         : "=&s"(saved_exec) : "v"(dst), "v"(dx), "v"(out), "v"(pflag) : "vcc", "memory");
        }
    }
}

__global__ __launch_bounds__(256) void invert_perm_kernel(const int32_t *__restrict__ perm, int64_t n,
                                                          int32_t *__restrict__ inv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) inv[perm[i]] = (int32_t)i;
}

__global__ __launch_bounds__(256) void cols_to_positions_kernel(const int32_t *__restrict__ col,
                                                                const int32_t *__restrict__ inv,
                                                                int64_t nnz, int32_t *__restrict__ pcol) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnz; k += stride) pcol[k] = inv[col[k]];
}


// Few, wide levels (multi-colour orderings: 2-16 levels of 10^5..10^7 rows):
// one plain launch per level, one lane per row, no flags and no polling -- the
// kernel boundary is the hand-off.  x may alias b.
template <typename RP>
__global__ __launch_bounds__(256) void trsv_level_kernel(const RP *__restrict__ row_ptr,
                                                         const int32_t *__restrict__ col,
                                                         const double *__restrict__ val,
                                                         const int32_t *__restrict__ perm,
                                                         int64_t begin, int64_t end,
                                                         const double *__restrict__ D, const double *b,
                                                         double *x, const int *stop) {
    if (stop && stop[1]) return;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = begin + (int64_t)blockIdx.x * 256 + threadIdx.x; i < end; i += stride) {
        const int r = perm[i];
        double acc = 0.0;
        for (int64_t k = (int64_t)row_ptr[r]; k < (int64_t)row_ptr[r + 1]; ++k)
            acc = fma(val[k], x[col[k]], acc);
        x[r] = (b[r] - acc) / D[r];
    }
}
constexpr int kFewLevels = 64;

// Host analysis: levels and the level-sorted permutation.
template <typename RP>
void build_perm(const RP *rp, const int32_t *col, int64_t n, bool backward,
                std::vector<int32_t> &perm, int &n_levels, int64_t &max_width,
                std::vector<int64_t> &level_ptr) {
    std::vector<int32_t> level(n, 0);
    int maxl = 0;
    if (!backward) {
        for (int64_t r = 0; r < n; ++r) {
            int l = 0;
            for (int64_t k = rp[r]; k < rp[r + 1]; ++k) l = std::max(l, level[col[k]] + 1);
            level[r] = l;
            maxl = std::max(maxl, l);
        }
    } else {
        for (int64_t r = n - 1; r >= 0; --r) {
            int l = 0;
            for (int64_t k = rp[r]; k < rp[r + 1]; ++k) l = std::max(l, level[col[k]] + 1);
            level[r] = l;
            maxl = std::max(maxl, l);
        }
    }
    n_levels = n ? maxl + 1 : 0;
    std::vector<int64_t> start(n_levels + 1, 0);
    for (int64_t r = 0; r < n; ++r) start[level[r] + 1]++;
    for (int l = 0; l < n_levels; ++l) start[l + 1] += start[l];
    perm.resize(n);
    level_ptr.assign(start.begin(), start.end());
    max_width = 0;
    for (int l = 0; l < n_levels; ++l) max_width = std::max<int64_t>(max_width, start[l + 1] - start[l]);
    for (int64_t r = 0; r < n; ++r) perm[start[level[r]]++] = (int32_t)r;
}

// Validates the triangular structure the solve relies on.
template <typename RP>
bool check_triangular(const RP *rp, const int32_t *col, int64_t n, bool backward) {
    for (int64_t r = 0; r < n; ++r)
        for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
            if (col[k] < 0 || col[k] >= n) return false;
            if (!backward && col[k] >= r) return false;
            if (backward && col[k] <= r) return false;
        }
    return true;
}

bis_status get_plan(bis_ctx *ctx, const bis_mat *T, bool backward, bis_trsv_plan **out) {
    bis_mat *M = const_cast<bis_mat *>(T);
    bis_trsv_plan *&slot = backward ? M->plan_bwd : M->plan_fwd;
    if (slot) { *out = slot; return BIS_OK; }
    const int64_t n = T->n_rows;
    bis_status st = BIS_OK;
    std::vector<int32_t> perm;
    bis_trsv_plan *p = new bis_trsv_plan;
    p->n = n;
    hipError_t e = hipMalloc(&p->perm, sizeof(int32_t) * (size_t)std::max<int64_t>(n, 1));
    // (xs, the level-scheduled kernels' scratch, is allocated at their first sweep: a matrix that ends up on the chained or the
    // tiled sweep never needs it)
    if (e == hipSuccess) e = hipMalloc(&p->ticket, sizeof(unsigned) * 4);
    if (e != hipSuccess) {
        ctx->err = std::string("sptrsv plan: ") + hipGetErrorString(e);
        bis_trsv_plan_destroy(p);
        return BIS_ERR_HIP;
    }
    const char *not_tri = backward ? "bis_bsptrsv: matrix is not strictly upper triangular"
                                   : "bis_sptrsv: matrix is not strictly lower triangular";
    bool blocks_done = false;
    if (bis_opts().trsv_host_analysis <= 0 && n > 0 && (int64_t)T->chunk_nnz + T->max_row_nnz + 8 <= 8192) {
        // a colour-sorted matrix decomposes into a few contiguous blocks of mutually independent rows:
        // sweep each with the streaming SpMV kernel + triangular epilogue (no levels, no row list needed)
        std::vector<int64_t> bounds;
        bool triangular = true;
        st = bis_trsv_blocks_device(ctx, T, backward, kFewLevels, bounds, backward ? nullptr : p->perm, triangular);
        if (st == BIS_OK && !triangular) { ctx->err = not_tri; st = BIS_ERR_INVALID; }
        if (st != BIS_OK) { bis_trsv_plan_destroy(p); return st; }
        if (!bounds.empty()) {
            const int nb = (int)bounds.size() - 1;
            for (int l = 0; l < nb && st == BIS_OK; ++l) {
                const int64_t r0 = backward ? bounds[l + 1] : bounds[l], r1 = backward ? bounds[l] : bounds[l + 1];
                bis_mat *v = nullptr;
                st = bis_mat_row_view(ctx, T, r0, r1, &v);
                if (st == BIS_OK) { p->level_views.push_back(v); p->level_row0.push_back(r0); }
            }
            if (st != BIS_OK) { bis_trsv_plan_destroy(p); return st; }
            p->n_levels = nb;
            p->max_level_width = 0;
            for (int l = 0; l < nb; ++l) p->max_level_width = std::max<int64_t>(p->max_level_width, std::llabs(bounds[l + 1] - bounds[l]));
            if (!backward) p->level_ptr = bounds; // forward: the blocks are levels of the identity row list (ILU(0) uses them)
            blocks_done = true;
        }
    }
    if (blocks_done) {
        slot = p;
        *out = p;
        return BIS_OK;
    }
    if (bis_opts().trsv_host_analysis <= 0) {
        // levels, level-sorted rows and the structure check on the device (bis_analysis.hip)
        bool triangular = true;
        st = bis_trsv_analyse_device(ctx, T, backward, p->perm, p->level_ptr, p->n_levels, p->max_level_width, triangular,
                                     bis_opts().trsv_chain != 0 ? &p->level : nullptr);
        if (st == BIS_OK && !triangular) { ctx->err = not_tri; st = BIS_ERR_INVALID; }
        if (st != BIS_OK) { bis_trsv_plan_destroy(p); return st; }
        if (p->n_levels <= kFewLevels && n > 0) { // the contiguity test below reads the permutation
            perm.resize(n);
            e = hipMemcpyAsync(perm.data(), p->perm, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
    } else { // host version: download the pattern and walk it serially
        std::vector<int64_t> rp(n + 1);
        std::vector<int32_t> col((size_t)std::max<int64_t>(T->nnz, 1));
        st = bis_mat_download(ctx, T, rp.data(), col.data(), nullptr);
        if (st == BIS_OK && !check_triangular(rp.data(), col.data(), n, backward)) { ctx->err = not_tri; st = BIS_ERR_INVALID; }
        if (st != BIS_OK) { bis_trsv_plan_destroy(p); return st; }
        build_perm(rp.data(), col.data(), n, backward, perm, p->n_levels, p->max_level_width, p->level_ptr);
        if (n)
            e = hipMemcpyAsync(p->perm, perm.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    if (e != hipSuccess) {
        ctx->err = std::string("sptrsv plan: ") + hipGetErrorString(e);
        bis_trsv_plan_destroy(p);
        return BIS_ERR_HIP;
    }
    // Few levels, each a contiguous ascending row range (a colour-sorted matrix):
    // sweep each level with the streaming SpMV kernel + triangular epilogue.
    if (p->n_levels <= kFewLevels && n > 0) {
        bool contiguous = true;
        for (int l = 0; l < p->n_levels && contiguous; ++l)
            for (int64_t i = p->level_ptr[l]; i + 1 < p->level_ptr[l + 1]; ++i)
                if (perm[i + 1] != perm[i] + 1) { contiguous = false; break; }
        if (contiguous && (int64_t)T->chunk_nnz + T->max_row_nnz + 8 <= 8192) {
            for (int l = 0; l < p->n_levels; ++l) {
                const int64_t r0 = perm[p->level_ptr[l]];
                bis_mat *v = nullptr;
                st = bis_mat_row_view(ctx, T, r0, r0 + (p->level_ptr[l + 1] - p->level_ptr[l]), &v);
                if (st != BIS_OK) { for (auto *m : p->level_views) bis_mat_destroy(ctx, m); p->level_views.clear(); break; }
                p->level_views.push_back(v);
                p->level_row0.push_back(r0);
            }
        }
    }
    slot = p;
    *out = p;
    return BIS_OK;
}

bis_status trsv_solve_impl(bis_ctx *ctx, const bis_mat *T, bool backward, double *x, const double *D,
                           const double *b, const char *&kernel) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, T && (T->n_rows == 0 || (x && D && b)), "sptrsv: bad arguments");
    BIS_REQUIRE(ctx, T->n_rows == T->n_cols, "sptrsv: square matrix required");
    const int64_t n = T->n_rows;
    if (n == 0) return BIS_OK;
    // natural orderings on a grid: the tiled sweep (modes: bis_trsv_tiled_build), tried first -- its plan
    // checks the dependency order itself, so where it applies the level analysis below is never made
    const bool tiled_allowed = bis_opts().trsv_tiled != 0 && bis_opts().trsv_inject_loss <= 0 && bis_opts().trsv_one_xcd <= 0;
    auto tiled_sweep = [&](bool *done) -> bis_status {
        *done = false;
        bis_mat *M = const_cast<bis_mat *>(T);
        bis_trsv_tiled *&ts = backward ? M->tiled_bwd : M->tiled_fwd;
        bool &tried = backward ? M->tiled_tried_bwd : M->tiled_tried_fwd;
        if (!tried) {
            tried = true;
            const bis_status tst = bis_trsv_tiled_build(ctx, T, backward, &ts);
            if (tst != BIS_OK) return tst;
        }
        if (!ts) return BIS_OK;
        *done = true;
        kernel = "trsv_tiled_kernel";
        return bis_trsv_tiled_solve(ctx, ts, x, D, b);
    };
    // ... except on grids with several unknowns per node and up to ~a million rows, where the chained sweep is tried first: the
    // unknowns of a node are a chain, the tiles of such a matrix are 2 x 2 x 2 nodes, and the chained sweep measured faster
    // (FEM-like 20^3 / 40^3 / 60^3 x 3: 0.40 / 0.84 / 1.42 ms against 0.53 / 1.05 / 1.63 tiled; 80^3 x 3 a tie at 2.3; 100^3 x 3
    // 4.8 / 6.4 against 3.0 -- there the wavefront is wider than the resident wave pairs).  The tiled sweep stays the fall-back.
    // The upper bound scales with the wave pairs the device keeps resident for this context: the wavefront of a grid of n rows is
    // ~n^(2/3) rows wide, so the rows a given number of pairs can follow grow with pairs^(3/2) (1.2 M rows measured at 2560 pairs).
    const double pairs_rel = (double)bis_trsv_chain_resident_pairs(ctx, T->rp64 != 0, backward) / 2560.0;
    const int64_t chain_first_max = (int64_t)(1200000.0 * pairs_rel * std::sqrt(pairs_rel));
    const bool chain_first = tiled_allowed && bis_opts().trsv_tiled < 0 && bis_opts().trsv_chain < 0 && T->grid[0] > 0 && T->grid[3] > 1 &&
                             n >= 16384 && n <= chain_first_max;
    if (tiled_allowed && !chain_first) {
        bool done = false;
        const bis_status tst = tiled_sweep(&done);
        if (tst != BIS_OK || done) return tst;
    }
    bis_trsv_plan *p = nullptr;
    bis_status st = get_plan(ctx, T, backward, &p);
    if (st != BIS_OK) return st;
    if (!p->level_views.empty()) {
        kernel = "spmv_rowblock_kernel (triangular epilogue, a launch per independent row block)";
        for (int l = 0; l < p->n_levels; ++l) {
            const int64_t r0 = p->level_row0[l];
            st = bis_spmv_trsv_level(ctx, p->level_views[l], x, x + r0, b + r0, D + r0);
            if (st != BIS_OK) return st;
        }
        return BIS_OK;
    }
    if (p->n_levels <= kFewLevels) {
        kernel = "trsv_level_kernel (a launch per level)";
        for (int l = 0; l < p->n_levels; ++l) {
            const int64_t lo = p->level_ptr[l], hi = p->level_ptr[l + 1];
            const int grid = (int)std::min<int64_t>((hi - lo + 255) / 256, (int64_t)ctx->n_cus * 32);
            if (T->rp64)
                hipLaunchKernelGGL(trsv_level_kernel<int64_t>, dim3(grid), dim3(256), 0, ctx->stream,
                                   (const int64_t *)T->row_ptr, T->col, T->val, p->perm, lo, hi, D, b, x, ctx->spmv_stop);
            else
                hipLaunchKernelGGL(trsv_level_kernel<int32_t>, dim3(grid), dim3(256), 0, ctx->stream,
                                   (const int32_t *)T->row_ptr, T->col, T->val, p->perm, lo, hi, D, b, x, ctx->spmv_stop);
        }
        BIS_HIP_CHECK(ctx, hipGetLastError());
        return BIS_OK;
    }
    // many narrow levels on a matrix without a grid: the chained sweep (bis_trsv_chain.hip) where its plan applies
    if (bis_opts().trsv_chain != 0 && bis_opts().trsv_inject_loss <= 0 && bis_opts().trsv_one_xcd <= 0) {
        bis_mat *M = const_cast<bis_mat *>(T);
        bis_trsv_chain *&cs = backward ? M->chain_bwd : M->chain_fwd;
        bool &tried = backward ? M->chain_tried_bwd : M->chain_tried_fwd;
        if (!tried) {
            tried = true;
            if (p->level) {
                const bis_status cst = bis_trsv_chain_build(ctx, T, backward, p->level, p->n_levels, &cs);
                hipFree(p->level);
                p->level = nullptr;
                if (cst != BIS_OK) return cst;
            }
        } else if (p->level) { // (a level plan rebuilt after the values changed: the chains depend on the pattern only)
            hipFree(p->level);
            p->level = nullptr;
        }
        if (cs) { kernel = "trsv_chain_kernel"; return bis_trsv_chain_solve(ctx, T, cs, x, D, b); }
    }
    if (chain_first) { // (no chained plan for this matrix after all)
        bool done = false;
        const bis_status tst = tiled_sweep(&done);
        if (tst == BIS_OK && done) { // the tiled sweep serves this triangle from now on: the level plan made for the chained attempt goes
            bis_mat *M = const_cast<bis_mat *>(T);
            bis_trsv_plan *&slot = backward ? M->plan_bwd : M->plan_fwd;
            bis_trsv_plan_destroy(slot);
            slot = nullptr;
        }
        if (tst != BIS_OK || done) return tst;
    }
    if (!p->xs) {
        const hipError_t xe = hipMalloc(&p->xs, sizeof(double) * (size_t)(n + 1)); // + one slot nobody publishes (test hook)
        if (xe != hipSuccess) { (void)hipGetLastError(); ctx->err = "sptrsv: out of memory for the scratch vector"; return BIS_ERR_HIP; }
    }
    const int fill_grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(fill_sentinel_kernel, dim3(fill_grid), dim3(256), 0, ctx->stream,
                       (unsigned long long *)p->xs, n + 1, p->ticket);
    // persistent grid: resident by construction (<= 8 workgroups of 256 per CU, 51 VGPRs)
    const int64_t n_tickets = (n + kTrsvT - 1) / kTrsvT;
    // Only ~one level is runnable at a time: keep a few of the widest levels
    // resident, at most one workgroup per CU -- idle pollers slow the hand-offs
    // (measured: HPCG-128 4.6 ms at 64-128 workgroups, 9.3 ms at 512, 22 ms at
    // 1024; Anderson-256 4.5 ms at 256, 13.5 ms at 1024).
    int64_t want = (4 * p->max_level_width + kTrsvT - 1) / kTrsvT + 1;
    const int one_xcd = bis_opts().trsv_one_xcd < 0 ? 0 : bis_opts().trsv_one_xcd;
    if (one_xcd) want = std::min<int64_t>(want, (int64_t)(ctx->n_cus / 8) * one_xcd) * 8; // per XCD x 8 XCDs
    else want = std::min<int64_t>(want, ctx->n_cus);
    if (bis_opts().trsv_grid > 0) want = bis_opts().trsv_grid;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(one_xcd ? n_tickets * 8 : n_tickets, want),
                                                                 (int64_t)ctx->n_cus * 8));
    // dependencies loaded per round trip: the whole row when it fits (HPCG-128 U, 13 per row: 3.05 ms with 16,
    // 5.28 ms with 8; Anderson-256, 3 per row: 2.4 ms with 4 or 8)
    const int batch = bis_opts().trsv_batch > 0 ? bis_opts().trsv_batch : (T->max_row_nnz > 8 ? 16 : T->max_row_nnz > 4 ? 8 : 4);
    // The scratch vector lives in LEVEL order (position in perm), so the polls and the
    // stores of neighbouring lanes fall into the same cache lines; pcol = positions
    // of the columns, built at the first solve.
    const int by_pos = bis_opts().trsv_by_pos < 0 ? 1 : bis_opts().trsv_by_pos;
    if (by_pos && !p->pcol && !p->no_pos && T->nnz > 0) {
        int32_t *inv = nullptr;
        BIS_HIP_CHECK(ctx, hipMalloc(&inv, sizeof(int32_t) * (size_t)n));
        hipError_t pe = hipMalloc(&p->pcol, sizeof(int32_t) * (size_t)T->nnz);
        if (pe != hipSuccess) { hipFree(inv); p->pcol = nullptr; ctx->err = "sptrsv: out of memory for the position table"; return BIS_ERR_HIP; }
        hipLaunchKernelGGL(invert_perm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           p->perm, n, inv);
        // the view's first non-zero: row views share the parent's arrays
        int64_t k0 = 0;
        { int64_t a64 = 0; int32_t a32 = 0;
          BIS_HIP_CHECK(ctx, hipMemcpyAsync(T->rp64 ? (void *)&a64 : (void *)&a32, T->row_ptr, T->rp64 ? 8 : 4, hipMemcpyDeviceToHost, ctx->stream));
          BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
          k0 = T->rp64 ? a64 : a32; }
        if (k0 != 0) { // a row view (absolute indices into the parent's arrays): keep the row-order scratch
            hipFree(p->pcol); p->pcol = nullptr; p->no_pos = true;
        } else
        hipLaunchKernelGGL(cols_to_positions_kernel, dim3((unsigned)std::min<int64_t>((T->nnz + 255) / 256, 8192)), dim3(256), 0,
                           ctx->stream, T->col, inv, T->nnz, p->pcol);
        BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        hipFree(inv);
    }
    const int32_t *dep = (by_pos && p->pcol) ? p->pcol : T->col;
    const int pos_flag = (by_pos && p->pcol) ? 1 : 0;
    // Test hook (bis_set_option("trsv_inject_loss", k)): the first dependency of the k-th non-zero's row is
    // redirected, for this one sweep, to the scratch slot nobody publishes -- the row gives up after
    // kSpinLimit polls, publishes NaN and raises the fault word.
    struct Restore { bis_ctx *c; int32_t *at; int32_t old; ~Restore() { if (at) { hipMemcpyAsync(at, &old, 4, hipMemcpyHostToDevice, c->stream); hipStreamSynchronize(c->stream); } } } restore{ctx, nullptr, 0};
    if (bis_opts().trsv_inject_loss > 0 && pos_flag && (int64_t)bis_opts().trsv_inject_loss <= T->nnz) {
        int32_t *at = p->pcol + (bis_opts().trsv_inject_loss - 1);
        const int32_t lost = (int32_t)n;
        BIS_HIP_CHECK(ctx, hipMemcpyAsync(&restore.old, at, 4, hipMemcpyDeviceToHost, ctx->stream));
        BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        BIS_HIP_CHECK(ctx, hipMemcpyAsync(at, &lost, 4, hipMemcpyHostToDevice, ctx->stream));
        restore.at = at;
        bis_opts().trsv_inject_loss = -1; // one shot
    }
    // rows of more than 8 dependencies: one wave per row (config-5 stand-in, ~35 per row: 10.8 / 25.3 ms per
    // forward / backward sweep with a lane per row -> 6.0 / 6.0 ms; HPCG-128, 13 per row: 2.88 / 3.05 -> 2.64 / 2.66;
    // Anderson-256, 3 per row and 22 K rows per level: 2.4 ms with a lane per row, 8.6 ms with a wave per row)
    // ... unless the levels are much wider than the resident waves (HPCG-512: 37 K rows per level against 4096
    // waves: 98 ms per sweep with a wave per row) and the rows still fit two lane batches
    const int64_t avg_width = p->n_levels > 0 ? n / p->n_levels : n;
    const int wave_auto = T->max_row_nnz > 16 || (T->max_row_nnz > 8 && avg_width <= (int64_t)ctx->n_cus * 16);
    // Where the rule says "a wave per row" the other kernel is sometimes the faster one (`unstr:80,80,80` as generated, 135 levels of
    // 11 K rows, 33 entries per row: forward 1.50 ms with a wave per row, 1.21 with a lane per row; backward 1.52 against 3.55), and
    // nothing known before the sweep tells the cases apart: the first sweep of a large triangle runs both (twice each, the second
    // run timed; the results are the same bits) and the plan keeps the faster one.  Not inside a device schedule (a stopped
    // schedule's launches are no-ops), not when x aliases an input (the sweep could not be repeated).
    if (p->wave_choice == 0 && bis_opts().trsv_wave < 0) {
        // (levels of 1024 rows or more only: on narrow ones a lane per row is hopeless -- `unstr:80,80,80` RCM-ordered, 215 rows per level: 105 ms
        // against 13 with a wave per row -- and the trial itself would cost more than it can win)
        const bool can_try = bis_opts().trsv_trial != 0 && wave_auto && !one_xcd && n >= 200000 && T->max_row_nnz <= 128 && avg_width >= 1024 &&
                             x != b && x != D && !ctx->spmv_stop && bis_opts().trsv_inject_loss <= 0 && bis_opts().trsv_grid <= 0;
        if (!can_try) {
            if (ctx->spmv_stop || x == b || x == D) { /* decide at a later sweep */ }
            else p->wave_choice = wave_auto ? 1 : 2;
        } else {
            hipEvent_t ev[2] = {nullptr, nullptr};
            bool good = hipEventCreate(&ev[0]) == hipSuccess && hipEventCreate(&ev[1]) == hipSuccess;
            bis_status tst = BIS_OK;
            for (int v = 1; v <= 2 && good && tst == BIS_OK; ++v) {
                p->wave_choice = v;
                for (int rep = 0; rep < 2 && tst == BIS_OK; ++rep) {
                    if (rep == 1) good = good && hipEventRecord(ev[0], ctx->stream) == hipSuccess;
                    tst = trsv_solve_impl(ctx, T, backward, x, D, b, kernel);
                }
                good = good && hipEventRecord(ev[1], ctx->stream) == hipSuccess && hipEventSynchronize(ev[1]) == hipSuccess &&
                       hipEventElapsedTime(&p->trial_ms[v - 1], ev[0], ev[1]) == hipSuccess;
            }
            if (ev[0]) hipEventDestroy(ev[0]);
            if (ev[1]) hipEventDestroy(ev[1]);
            (void)hipGetLastError();
            p->wave_choice = (good && tst == BIS_OK && p->trial_ms[1] < 0.9f * p->trial_ms[0]) ? 2 : 1;
            kernel = p->wave_choice == 1 ? "sptrsv_wave_kernel" : "sptrsv_syncfree_kernel";
            return tst; // (x holds the solution: the last of the four sweeps)
        }
    }
    const int wave_mode = bis_opts().trsv_wave >= 0 ? bis_opts().trsv_wave : (p->wave_choice ? (p->wave_choice == 1 ? 1 : 0) : (wave_auto ? 1 : 0));
    if (wave_mode && !one_xcd) {
        // a few levels of rows in flight, one row per wave; at most 4 workgroups per CU: every wave of the
        // grid must be resident (static round robin, see the kernel)
        // Rows are dealt to the waves round robin, so EVERY wave of the grid must be resident: the grid is
        // capped by the occupancy the runtime reports for this kernel (the launch bound guarantees
        // kWaveBlocksPerCU), whatever trsv_grid asks for.  If the device is shared and some workgroup still
        // cannot start, its rows are never published: the waiting rows give up after kSpinLimit polls and
        // the sweep fails with BIS_ERR_SYNC at the next blocking call instead of returning NaNs silently.
        static int resident[2] = {0, 0};
        int &res = resident[T->rp64 ? 1 : 0];
        if (res == 0) {
            int nb = 0;
            hipError_t oe = T->rp64 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sptrsv_wave_kernel<int64_t>, kTrsvT, 0)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sptrsv_wave_kernel<int32_t>, kTrsvT, 0);
            res = (oe == hipSuccess && nb > 0) ? std::min(nb, 8) : 1; // (what the runtime reports: 8 at 30 VGPRs; the launch bound guarantees 4)
            (void)hipGetLastError();
        }
        // workgroups per CU: 4 on narrow levels (more idle pollers slow the hand-offs), 6 on levels of >= 1024 rows, where the rows in
        // flight are what counts (`unstr:80,80,80` as generated, 11 K rows per level: 2.51 / 1.49 / 1.27 / 1.38 ms at 2 / 4 / 6 / 8);
        // option trsv_wave_wgs: up to what the runtime reports resident (a grid that needs all 8 slots of every CU does not always get
        // them -- see bis_ilu0.hip --: 6 leaves a quarter of the slots spare)
        const int per_cu = std::min(res, bis_opts().trsv_wave_wgs > 0 ? bis_opts().trsv_wave_wgs : (avg_width >= 1024 ? 6 : kWaveBlocksPerCU));
        int64_t wg = (4 * p->max_level_width + 3) / 4 + 1;
        if (bis_opts().trsv_grid > 0) wg = bis_opts().trsv_grid;
        // option "device_share" = k: k processes run sweeps on this device at the same time (ranks of a test or a rehearsal
        // sharing one GPU), each keeps to 1/k of the residency so that all their grids fit together
        const int share = std::max(1, bis_opts().device_share);
        wg = std::max<int64_t>(1, std::min<int64_t>(wg, std::min<int64_t>((n + 3) / 4, std::max<int64_t>(1, (int64_t)ctx->n_cus * per_cu / share))));
        if (T->rp64)
            hipLaunchKernelGGL(sptrsv_wave_kernel<int64_t>, dim3((unsigned)wg), dim3(kTrsvT), 0, ctx->stream,
                               (const int64_t *)T->row_ptr, dep, T->val, p->perm, n, D, b, x,
                               (unsigned long long *)p->xs, p->ticket, pos_flag, ctx->fault_dev, ctx->spmv_stop);
        else
            hipLaunchKernelGGL(sptrsv_wave_kernel<int32_t>, dim3((unsigned)wg), dim3(kTrsvT), 0, ctx->stream,
                               (const int32_t *)T->row_ptr, dep, T->val, p->perm, n, D, b, x,
                               (unsigned long long *)p->xs, p->ticket, pos_flag, ctx->fault_dev, ctx->spmv_stop);
        BIS_HIP_CHECK(ctx, hipGetLastError());
        kernel = "sptrsv_wave_kernel";
        return BIS_OK;
    }
    kernel = "sptrsv_syncfree_kernel";
#define BIS_TRSV_LAUNCH(RP, ONE, B)                                                                    \
    hipLaunchKernelGGL((sptrsv_syncfree_kernel<RP, ONE, B>), dim3(grid), dim3(kTrsvT), 0, ctx->stream, \
                       (const RP *)T->row_ptr, dep, T->val, p->perm, n, D, b, x,                      \
                       (unsigned long long *)p->xs, p->ticket, pos_flag, ctx->fault_dev, ctx->spmv_stop)
#define BIS_TRSV_B(RP, ONE)                                                                            \
    do {                                                                                               \
        if (batch >= 32) BIS_TRSV_LAUNCH(RP, ONE, 32);                                                 \
        else if (batch >= 16) BIS_TRSV_LAUNCH(RP, ONE, 16);                                            \
        else if (batch >= 8) BIS_TRSV_LAUNCH(RP, ONE, 8);                                              \
        else BIS_TRSV_LAUNCH(RP, ONE, 4);                                                              \
    } while (0)
    if (T->rp64) { if (one_xcd) BIS_TRSV_B(int64_t, true); else BIS_TRSV_B(int64_t, false); }
    else { if (one_xcd) BIS_TRSV_B(int32_t, true); else BIS_TRSV_B(int32_t, false); }
#undef BIS_TRSV_B
#undef BIS_TRSV_LAUNCH
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}

// one sweep = every launch of the call (sentinel fill + the sweep kernel, or a launch per level): HIP-event bracketed on the
// context's stream while bis_profile_enable is on (bis_profile_read_sweeps)
bis_status trsv_solve(bis_ctx *ctx, const bis_mat *T, bool backward, double *x, const double *D, const double *b) {
    BIS_CTX_OK(ctx);
    const bool prof = ctx->profile && T && T->n_rows > 0;
    if (prof) {
        if (ctx->prof_sweep_used == ctx->prof_sweep_events.size()) {
            hipEvent_t a, e;
            hipEventCreate(&a);
            hipEventCreate(&e);
            ctx->prof_sweep_events.emplace_back(a, e);
        }
        hipEventRecord(ctx->prof_sweep_events[ctx->prof_sweep_used].first, ctx->stream);
    }
    const char *kernel = "";
    const bis_status st = trsv_solve_impl(ctx, T, backward, x, D, b, kernel);
    if (prof) {
        hipEventRecord(ctx->prof_sweep_events[ctx->prof_sweep_used].second, ctx->stream);
        ++ctx->prof_sweep_used;
    }
    if (T && st == BIS_OK && kernel[0]) const_cast<bis_mat *>(T)->sweep_kernel[backward ? 1 : 0] = kernel;
    return st;
}

} // namespace

bis_status bis_trsv_level_sets(bis_ctx *ctx, const bis_mat *T_lower, const std::vector<int64_t> **level_ptr,
                               const int32_t **perm_dev) {
    bis_trsv_plan *p = nullptr;
    bis_status st = get_plan(ctx, T_lower, false, &p);
    if (st != BIS_OK) return st;
    *level_ptr = &p->level_ptr;
    *perm_dev = p->perm;
    return BIS_OK;
}

extern "C" {

const char *bis_mat_sweep_kernel(const bis_mat *T, int backward) { return T ? T->sweep_kernel[backward ? 1 : 0] : ""; }

bis_status bis_sptrsv(bis_ctx *ctx, const bis_mat *L_strict, double *x, const double *D,
                      const double *b) {
    return trsv_solve(ctx, L_strict, false, x, D, b);
}

bis_status bis_bsptrsv(bis_ctx *ctx, const bis_mat *U_strict, double *x, const double *D,
                       const double *b) {
    return trsv_solve(ctx, U_strict, true, x, D, b);
}

// two_stage_gauss_seidel, kernels.hpp:312-333.
bis_status bis_two_stage_gauss_seidel(bis_ctx *ctx, const bis_mat *strict, double *tmp,
                                      double *work, const double *D_inv, const double *input,
                                      double *output, int64_t n, int inner_iters) {
    BIS_CTX_OK(ctx);
    bis_status st = bis_elemwise_mult_vectors(ctx, work, D_inv, input, n, 1.0);   // :317
    if (st == BIS_OK) st = bis_copy_vector(ctx, output, work, n);                  // :319
    for (int inner = 1; st == BIS_OK && inner <= inner_iters; ++inner) {
        st = bis_spmv(ctx, strict, work, tmp);                                     // :323
        if (st == BIS_OK) st = bis_elemwise_mult_vectors(ctx, tmp, D_inv, tmp, n, -1.0); // :325
        std::swap(work, tmp);                                                      // :327
        if (st == BIS_OK) st = bis_sum_vectors(ctx, output, output, work, n, 1.0); // :331
    }
    return st;
}

// apply_preconditioner, kernels.hpp:336-414.
bis_status bis_apply_preconditioner(bis_ctx *ctx, int pc, int64_t n, const bis_mat *L_strict,
                                    const bis_mat *U_strict, const double *A_D,
                                    const double *A_D_inv, const double *L_D, const double *U_D,
                                    double *output, double *input, double *tmp, double *work,
                                    int outer_iters, int inner_iters) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, n >= 0 && outer_iters >= 1, "bis_apply_preconditioner: bad arguments");
    double *input_storage = nullptr;
    bis_status st = BIS_OK;
    if (outer_iters > 1) { // :348-352 (the one place the reference allocates in a kernel)
        st = bis_vec_alloc(ctx, n, &input_storage);
        if (st == BIS_OK) st = bis_copy_vector(ctx, input_storage, input, n);
    }
    for (int i = 0; st == BIS_OK && i < outer_iters; ++i) {
        switch (pc) {
        case BIS_PC_JACOBI:
            st = bis_elemwise_div_vectors(ctx, output, input, A_D, n, 1.0);        // :357
            break;
        case BIS_PC_GAUSS_SEIDEL:
            st = bis_sptrsv(ctx, L_strict, output, A_D, input);                    // :359
            break;
        case BIS_PC_BACKWARDS_GAUSS_SEIDEL:
            st = bis_bsptrsv(ctx, U_strict, output, A_D, input);                   // :361
            break;
        case BIS_PC_SYMMETRIC_GAUSS_SEIDEL:
            st = bis_sptrsv(ctx, L_strict, tmp, A_D, input);                       // :365
            if (st == BIS_OK) st = bis_elemwise_mult_vectors(ctx, tmp, tmp, A_D, n, 1.0); // :369
            if (st == BIS_OK) st = bis_bsptrsv(ctx, U_strict, output, A_D, tmp);   // :373
            break;
        case BIS_PC_TWO_STAGE_GS:
            st = bis_two_stage_gauss_seidel(ctx, L_strict, tmp, work, A_D_inv, input, output, n,
                                            inner_iters);                          // :376
            break;
        case BIS_PC_SYMMETRIC_TWO_STAGE_GS:
            st = bis_two_stage_gauss_seidel(ctx, L_strict, tmp, work, A_D_inv, input, output, n,
                                            inner_iters);                          // :379
            if (st == BIS_OK) st = bis_elemwise_mult_vectors(ctx, output, output, A_D, n, 1.0); // :382
            if (st == BIS_OK)
                st = bis_two_stage_gauss_seidel(ctx, U_strict, tmp, work, A_D_inv, output, output,
                                                n, inner_iters);                   // :384
            break;
        case BIS_PC_ILU0:
            st = bis_sptrsv(ctx, L_strict, tmp, L_D, input);                       // :390
            if (st == BIS_OK) st = bis_bsptrsv(ctx, U_strict, output, U_D, tmp);   // :394
            break;
        default:
            st = bis_copy_vector(ctx, output, input, n);                           // :398
        }
        if (st == BIS_OK && outer_iters > 1 && i != outer_iters - 1)
            st = bis_copy_vector(ctx, input, output, n);                           // :401-403
    }
    if (st == BIS_OK && outer_iters > 1) st = bis_copy_vector(ctx, input, input_storage, n); // :406-408
    if (input_storage) bis_vec_free(ctx, input_storage);
    return st;
}

} // extern "C"
