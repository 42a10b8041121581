// bis_trsv_tiled.hip -- natural-order sparse triangular sweeps (reference kernels.hpp:54-107,
// serial there) with the dependency hand-offs kept INSIDE a workgroup wherever possible.
//
// The level-scheduled sweep of bis_sptrsv.hip pays one cross-CU hand-off (store to memory, poll from
// memory: 2-3 us under load, profiles/r02_a_trsv_*) per dependency level -- 766 of them on a 256^3
// 7-point grid.  Here the rows are cut into TILES: intervals of B consecutive rows of the processing
// order (ascending rows for the forward, descending for the backward sweep; any linear extension of the
// dependency order would do).  One workgroup solves one tile:
//
//   * inside the tile the rows are sorted by their LOCAL dependency level (longest path through
//     in-tile dependencies) and cut into steps of at most 64 mutually independent rows; ONE compute
//     wave walks the steps, a lane per row, and accumulates acc = fma(val, x[col], acc) in CRS order --
//     the reference's arithmetic exactly -- taking in-tile operands from an LDS copy of the tile's
//     results (an LDS round trip per step, ~0.1 us, instead of a memory round trip);
//   * operands produced by EARLIER tiles are fetched by a poller wave: the tile's distinct external
//     columns are listed in first-need order, the poller spins on the producers' published results
//     ("the data is the flag": a sentinel-initialised scratch vector, one sc1 store per row, as in the
//     level-scheduled sweep) and drops them into LDS, where the compute wave finds them;
//   * two loader waves stream the tile's matrix entries (in step order, lane-major) and the per-row
//     operands (row index, length, b, D) into LDS rings ahead of the compute wave, so the compute wave
//     itself never waits for HBM.
//
// Tiles are taken by ticket in processing order by a persistent grid; every dependency of a tile lies in
// an earlier tile (the tiles are intervals of a linear extension), whose workgroup is resident or done:
// progress is guaranteed, every wait is bounded (a lost hand-off raises the context's fault word).
// The number of cross-CU hand-offs on the critical path drops from the number of levels to the number of
// tile boundaries a dependency chain crosses, and each of them overlaps with in-tile work.
#include "bis_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <type_traits>
#include <rocprim/rocprim.hpp>

struct bis_trsv_tiled {
    int64_t n = 0;
    int n_tiles = 0, max_rows = 0;
    int64_t n_steps = 0, n_quads = 0, n_ext = 0;
    // device arrays.  "slot" = position of a row in (tile, local level, processing order) order.
    int32_t *slot_row = nullptr;    // [n]  slot -> row
    int4 *step_desc = nullptr;      // per tile, per step: {first slot, rows | quads per row << 8, first quad, external ordinals first needed up to and incl. this step}, relative to the tile
    int64_t *tile_slot0 = nullptr;  // [n_tiles + 1]
    int64_t *tile_step0 = nullptr;  // [n_tiles + 1] index into step_desc
    int64_t *tile_quad0 = nullptr;  // [n_tiles + 1]
    int64_t *tile_ext0 = nullptr;   // [n_tiles + 1]
    int4 *quad_code = nullptr;      // [n_quads] 4 consecutive entries of one row: BYTE offsets of their operands in the tile's LDS operand array
    double2 *quad_val = nullptr;    // [2 n_quads] their values (padding: value 0, operand = the zero slot)
    int32_t *ext_src = nullptr;     // [n_ext]  slot (global) whose published result the external ordinal stands for
    unsigned long long *xs = nullptr; // [n + 1] published results, by slot
    unsigned *ticket = nullptr;
};

void bis_trsv_tiled_destroy(bis_trsv_tiled *p) {
    if (!p) return;
    hipFree(p->slot_row); hipFree(p->step_desc); hipFree(p->tile_slot0); hipFree(p->tile_step0); hipFree(p->tile_quad0);
    hipFree(p->tile_ext0); hipFree(p->quad_code); hipFree(p->quad_val); hipFree(p->ext_src);
    hipFree(p->xs); hipFree(p->ticket);
    delete p;
}

namespace {

constexpr unsigned long long kSentinel = 0x7FF85EA71E55C0DEull; // quiet NaN + payload (same as bis_sptrsv.hip)
constexpr unsigned long long kCanonNaN = 0x7FF8000000000000ull;
// LDS budget of a workgroup: 51 KiB (operand rings 16, quad ring 24.6, per-row rings 10), 3 workgroups per CU -- which is
// also what the kernel's ~135 VGPRs allow.  The quad ring is what the size buys: the entry loader has one round of
// global loads in flight at a time, so the ring's depth is how far it runs ahead of the compute wave.  Measured (sweeps
// of HPCG-256 / the 7-point 256^3 grid, ms): a 128-quad ring with 4 workgroups per CU 2.31 / 0.81, 256 quads with 3 per CU
// 2.17 / 0.69, 512 quads 2.14 / 0.66 -- and 1.75 with the 8x8x4 tiles the deeper ring makes affordable; 64 or 256 quads per
// loader round instead of 128, 256 external ordinals per poll round instead of 128: no better.
template <int OWN, int EXT, int WINDOW, int RINGQ, int RINGSLOT, int QCHUNK, int SCHUNK, int GROUP>
struct TiledCfg {
    static constexpr int kGroup = GROUP;     // quads of a row the compute wave reads together (registers: 20 per quad)
    static constexpr int kOwn = OWN;         // LDS ring of the tile's own results, by slot: an in-tile operand must have been
                                             // produced fewer than kOwn - 128 slots before its consumer, else it is fetched like an external one
    static constexpr int kExt = EXT;         // LDS ring of external operands, by ordinal (first-need order)
    static constexpr int kExtWindow = WINDOW; // an ordinal may be used again while it is among the last kExtWindow ones first needed
    static constexpr int kZeroSlot = OWN + EXT; // operand of padding entries: 0.0, so fma(0, 0, acc) leaves acc as it is
    static constexpr int kOpnd = OWN + EXT + 2;
    static constexpr int kRingQ = RINGQ;     // quad ring (48 B per quad)
    static constexpr int kRingSlot = RINGSLOT; // per-row operand ring (20 B per row)
    static constexpr int kQuadChunk = QCHUNK; // quads per loader round
    static constexpr int kSlotChunk = SCHUNK; // rows per loader round
};
using Cfg = TiledCfg<1024, 1024, 256, 512, 512, 128, 256, 4>;
constexpr int kMaxB = 32768;      // rows per tile at most
constexpr unsigned kSpinLds = 1u << 26;  // polls of an LDS word before a wave gives up (several seconds: longer than the poller's budget below)
constexpr unsigned kSpinMem = 1u << 20;  // polls of a memory word (about a second)

enum { C_TICKET = 0, C_Q_LOADED, C_Q_DONE, C_SLOT_LOADED, C_SLOT_DONE, C_EXT_WM, C_EXT_SAFE, C_N = 8 };

// Hand-offs between the waves of one workgroup go through LDS words.  The LDS executes one wave's
// operations in issue order, so "data writes, then the watermark write" needs no wait in between, and a
// reader that has SEEN the watermark reads the data behind it; all the code has to prevent is the compiler
// moving LDS accesses across the watermark access (the empty asm statements).  C++ release/acquire at
// workgroup scope would be correct too but also drains the wave's outstanding GLOBAL stores
// (s_waitcnt vmcnt(0): a memory round trip per step -- measured 25 ms instead of 1 ms per sweep).
__device__ __forceinline__ unsigned lds_acquire(const unsigned *p) {
    const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void lds_release(unsigned *p, unsigned v) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned long long lds_word(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef double v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int4 nt_load(const int4 *p) {
    const v4i_t v = __builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double2 nt_load(const double2 *p) {
    const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(p));
    return make_double2(v.x, v.y);
}

__global__ __launch_bounds__(256) void fill_sentinel_kernel(unsigned long long *xs, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) xs[i] = kSentinel;
}

// quad_val[j] = val[src[j]] (src < 0: padding)
__global__ __launch_bounds__(256) void gather_entries_kernel(const double *__restrict__ val, const int64_t *__restrict__ src,
                                                             int64_t n, double *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) {
        const int64_t k = src[j];
        out[j] = k >= 0 ? val[k] : 0.0;
    }
}

struct __attribute__((aligned(16))) QuadRec { int4 c; double2 va, vb; };

struct TiledArgs {
    const int32_t *slot_row;
    const int4 *step_desc;
    const int64_t *tile_slot0, *tile_step0, *tile_quad0, *tile_ext0;
    const int4 *quad_code;
    const double2 *quad_val;
    const int32_t *ext_src;
    unsigned long long *xs;
    unsigned *ticket;
    const double *D, *b;
    double *x;
    unsigned *fault;
    const int *stop; // a device schedule's stop flag (flags[1]): the sweep is a no-op once it is set
    int n_tiles;
    int backoff;    // option trsv_tile_backoff
    int exp_flags;  // option trsv_tile_exp (experiments, timing only -- results are wrong): 1 no x store, 2 no xs store, 4 no b / D loads,
                    // 8 no entry loads, 16 external operands taken as delivered, 32 tiles dealt statically instead of by ticket,
                    // 64 stores of a step unmasked (results right), 128 nothing (the experiment build itself)
    long long *dbg; // DBG only, 16 words per tile (8, 9: the poller's rounds and the core cycles they took): start, end (s_memrealtime, 100 MHz), core cycles the compute wave waited for the
                    // loaders / for external operands, end of the quad loader / slot loader / poller, steps
    long long *dbg_step; // DBG only: core-cycle stamps inside the steps of the middle tile, 8 per step
    long long *dbg_pub, *dbg_del; // DBG only: s_memrealtime at which a slot's result was published / an external ordinal was delivered
};

__device__ __forceinline__ int wave_min_int(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}

template <typename CFG, bool DBG, bool EXP = false, int kPollBlock = 128>
__global__ __launch_bounds__(256) void trsv_tiled_kernel(const TiledArgs a) {
    const int exp_flags = EXP ? a.exp_flags : 0; // (the timing experiments are a build of their own: nothing of them in the product's loops)
    constexpr int kOwn = CFG::kOwn, kExt = CFG::kExt, kExtWindow = CFG::kExtWindow, kZeroSlot = CFG::kZeroSlot, kOpnd = CFG::kOpnd;
    constexpr int kRingQ = CFG::kRingQ, kRingSlot = CFG::kRingSlot, kQuadChunk = CFG::kQuadChunk, kSlotChunk = CFG::kSlotChunk;
    // operands of the tile's rows, indexed directly by the entry codes (one LDS read per operand, no branch on where it
    // comes from): [0, kOwn) ring of the tile's own results by slot, [kOwn, kOwn + kExt) ring of the external operands
    // by ordinal, then the zero slot.
    __shared__ unsigned long long opnd[kOpnd];
    // quad ring: codes and values of a quad side by side (one index computation for the three reads; a 48-byte stride keeps
    // 16-byte reads of consecutive lanes off each other's banks, which the 32-byte stride of a values-only array does not)
    __shared__ QuadRec ring_q[kRingQ + 1];       // (+ one permanent quad of padding entries, see the compute wave)
    __shared__ int ring_row[kRingSlot];
    __shared__ double2 ring_bD[kRingSlot];
    __shared__ unsigned ctl[C_N];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (a.stop && a.stop[1]) return;
    if (threadIdx.x == 0) {
        opnd[kZeroSlot] = 0ull;
        ring_q[kRingQ].c = make_int4(8 * kZeroSlot, 8 * kZeroSlot, 8 * kZeroSlot, 8 * kZeroSlot);
        ring_q[kRingQ].va = make_double2(0.0, 0.0);
        ring_q[kRingQ].vb = make_double2(0.0, 0.0);
    }
    unsigned dealt = blockIdx.x;
    for (;;) {
        if (threadIdx.x == 0) {
            unsigned tt = (unsigned)a.n_tiles;
            if (exp_flags & 32) { tt = dealt; dealt += gridDim.x; }
            else tt = atomicAdd(a.ticket, 1u);
            ctl[C_TICKET] = tt;
        }
        if (threadIdx.x >= 1 && threadIdx.x < C_N) ctl[threadIdx.x] = threadIdx.x == C_EXT_SAFE ? (unsigned)(-kExtWindow) : 0u;
        __syncthreads();
        const int t = (int)ctl[C_TICKET];
        if (t >= a.n_tiles) return; // every wave reaches this once the tickets run out
        const int64_t slot0 = a.tile_slot0[t];
        const int n_slots = (int)(a.tile_slot0[t + 1] - slot0);
        const int64_t sd0 = a.tile_step0[t];
        const int n_steps = (int)(a.tile_step0[t + 1] - sd0);
        const int64_t quad0 = a.tile_quad0[t];
        const int n_quads = (int)(a.tile_quad0[t + 1] - quad0);
        const int64_t ext0 = a.tile_ext0[t];
        const int n_ext = (int)(a.tile_ext0[t + 1] - ext0);
        const long long t_start = DBG ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
        long long w_load = 0, w_ext = 0;
        if (DBG && lane == 0) { // where the wave sits: HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SE ...) and the XCD
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            a.dbg[(int64_t)t * 16 + 10 + wave] = (long long)hw | ((long long)(xcc & 0xf) << 32);
        }

        if (wave == 0) {
            __builtin_amdgcn_s_setprio(3); // the step chain is the sweep's critical path: issue before the helper waves that share the SIMD
            // ---- compute wave: one lane per row of the step, CRS-order fma chain.  What counts is the number of wave
            // instructions per step: no predication (lanes beyond the step's rows read valid LDS and are masked at the
            // stores only), padding entries multiply 0 by the zero slot, readiness is three watermarks (the two loaders',
            // the poller's) cached in registers and re-read only when they do not cover the step, the wave's own progress
            // is published every 8 steps.
            int4 d_cur = a.step_desc[sd0 + min(lane, n_steps - 1)];
            int4 p_cur = d_cur; // the following batch of 64 step descriptors, fetched a batch ahead
            int q_loaded = 0, slot_loaded = 0, ext_wm = 0; // cached watermarks
            int ext_prev = 0;
            const long long c_first = DBG ? (long long)__builtin_readcyclecounter() : 0; // (descriptors are here: the steps begin)
            auto opnd_at = [&](int byte_off) { return (const unsigned long long *)((const char *)opnd + byte_off); };
            auto stamp = [&](int s, int k) {
                if (DBG && t == a.n_tiles / 2 && s < 64) {
                    const long long c = (long long)__builtin_readcyclecounter();
                    if (lane == 0) a.dbg_step[s * 8 + k] = c;
                }
            };
            // The loop is instantiated per row length (quads per row) where all steps of the tile agree on it -- the common
            // case on a stencil -- and once for the general case: with the length a constant there is no branch over it and
            // no padding-quad select in the step (the step is ~110 wave instructions at 8-10 cycles each; that count, not
            // memory, is two thirds of the sweep: DESIGN.md section 4, second pass).
            auto run_steps = [&](auto nqt) {
            constexpr int NQT = decltype(nqt)::value; // 0: read per step
            // lane j: what step j needs loaded / delivered (at most 64 steps in these instances; lanes past the last step repeat it)
            [[maybe_unused]] const int my_w = d_cur.y & 0xff;
            [[maybe_unused]] const int my_qe = d_cur.z + my_w * NQT, my_se = d_cur.x + my_w, my_xe = d_cur.w;
            [[maybe_unused]] int ready_to = 0; // steps below it are covered by the cached watermarks
            for (int s = 0; s < n_steps; ++s) {
                stamp(s, 0);
                const int j = s & 63;
                if constexpr (NQT == 0) { // (the instances per row length have at most 64 steps: no second batch -- and a load nobody
                                          // consumes would leave the register allocator free to reuse its destination, which costs
                                          // an s_waitcnt vmcnt(0), i.e. the previous step's stores, in EVERY step: seen, 0.85 -> 1.09 ms)
                    if (__builtin_expect(j == 0, 0)) {
                        if (s > 0) d_cur = p_cur;
                        p_cur = a.step_desc[sd0 + min(s + 64 + lane, n_steps - 1)];
                    }
                }
                const int slot_b = __builtin_amdgcn_readlane(d_cur.x, j), wn = __builtin_amdgcn_readlane(d_cur.y, j);
                const int quad_b = __builtin_amdgcn_readlane(d_cur.z, j), ext_end = __builtin_amdgcn_readlane(d_cur.w, j);
                const int w = wn & 0xff, nq = NQT ? NQT : (wn >> 8);
                const int slot_e = slot_b + w, quad_e = quad_b + w * nq;
                // (a lone wave pays for every taken branch with a refill of its instruction buffer: the step's usual path falls through.
                // Where a lane holds each step's descriptor -- the instances per row length -- the three watermark tests are made for
                // all steps at once when the watermarks are re-read, and the step itself compares its number with the first uncovered one)
                const bool must_wait = (NQT && !(EXP && (exp_flags & 256))) ? s >= ready_to : (q_loaded < quad_e || slot_loaded < slot_e || ext_wm < ext_end);
                if (__builtin_expect(must_wait, 0)) {
                    // about to wait: tell the loaders and the poller how far their rings are free
                    if (lane == 0) {
                        lds_release(&ctl[C_Q_DONE], (unsigned)quad_b);
                        lds_release(&ctl[C_SLOT_DONE], (unsigned)slot_b);
                        lds_release(&ctl[C_EXT_SAFE], (unsigned)(ext_prev - kExtWindow));
                    }
                    unsigned spins = 0;
                    const long long t0 = DBG ? (long long)__builtin_readcyclecounter() : 0;
                    for (;;) {
                        q_loaded = __builtin_amdgcn_readfirstlane((int)lds_acquire(&ctl[C_Q_LOADED]));
                        slot_loaded = __builtin_amdgcn_readfirstlane((int)lds_acquire(&ctl[C_SLOT_LOADED]));
                        if (q_loaded >= quad_e && slot_loaded >= slot_e) break;
                        if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    const long long t1 = DBG ? (long long)__builtin_readcyclecounter() : 0;
                    if (DBG) w_load += t1 - t0;
                    spins = 0; // (a budget of its own: the poller, which always moves the watermark on, gives up long before it runs out)
                    for (;;) {
                        ext_wm = __builtin_amdgcn_readfirstlane((int)lds_acquire(&ctl[C_EXT_WM]));
                        if (ext_wm >= ext_end) break;
                        if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (DBG) w_ext += (long long)__builtin_readcyclecounter() - t1;
                    if constexpr (NQT != 0) {
                        const unsigned long long uncovered = __ballot(my_qe > q_loaded || my_se > slot_loaded || my_xe > ext_wm);
                        ready_to = max(s + 1, uncovered ? (int)__builtin_ctzll(uncovered) : 64);
                    }
                }
                stamp(s, 1);
                // lanes past the step's width take the last row's slots (loaded, in range); their results are masked at the store
                const int wl = min(lane, w - 1);
                const int sl = (slot_b + wl) & (kRingSlot - 1);
                const int row = ring_row[sl];
                const double2 bd = ring_bD[sl];
                double acc = 0.0;
                int qi = quad_b + wl;
                // A step is a chain of LDS round trips -- a quad's codes, then its operands -- with the step's whole duration
                // on the sweep's critical path (measured, tools/trsv_tile_debug.py: the sweep is the sum over the three tile
                // directions of tiles x (steps to the face x step time + hand-off)).  The quads of a row are therefore taken
                // G at a time: all their codes and values in ONE round trip, all their operands in a second one, then the
                // fma chain in CRS order as before.  Rows of fewer quads read the permanent padding quad (operand 0.0,
                // value 0.0: fma(0, 0, acc) as for the padding inside a quad).
                auto group = [&](auto gtag, int g_left) {
                    constexpr int G = decltype(gtag)::value;
                    int4 c[G];
                    double2 va[G], vb[G];
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const int qq = u < g_left ? ((qi + u * w) & (kRingQ - 1)) : kRingQ;
                        c[u] = ring_q[qq].c;
                        va[u] = ring_q[qq].va;
                        vb[u] = ring_q[qq].vb;
                    }
                    if (DBG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(s, 2); }
                    unsigned long long xo[G][4];
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        // (the codes are byte offsets: an LDS read with the array's base in its offset field, no shift)
                        xo[u][0] = lds_word(opnd_at(c[u].x)); xo[u][1] = lds_word(opnd_at(c[u].y));
                        xo[u][2] = lds_word(opnd_at(c[u].z)); xo[u][3] = lds_word(opnd_at(c[u].w));
                    }
                    if (DBG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(s, 3); }
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        acc = fma(va[u].x, __longlong_as_double((long long)xo[u][0]), acc);
                        acc = fma(va[u].y, __longlong_as_double((long long)xo[u][1]), acc);
                        acc = fma(vb[u].x, __longlong_as_double((long long)xo[u][2]), acc);
                        acc = fma(vb[u].y, __longlong_as_double((long long)xo[u][3]), acc);
                    }
                    qi += G * w;
                };
                if constexpr (NQT != 0) group(std::integral_constant<int, NQT>{}, NQT);
                else if (nq == 1) group(std::integral_constant<int, 1>{}, 1);
                else if (nq == 2) group(std::integral_constant<int, 2>{}, 2);
                else for (int g = 0; g < nq; g += CFG::kGroup) group(std::integral_constant<int, CFG::kGroup>{}, nq - g);
                if (DBG) { asm volatile("" :: "v"(acc)); stamp(s, 4); }
                const double res = (bd.x - acc) / bd.y;
                unsigned long long out = (unsigned long long)__double_as_longlong(res);
                if (DBG) { asm volatile("" :: "v"(out)); stamp(s, 5); }
                if (res != res) out = kCanonNaN; // never publish the sentinel pattern
                if (EXP && (exp_flags & 64)) { // (the stores unmasked: the lanes past w hold the last row's result again)
                    __hip_atomic_store(&opnd[(slot_b + wl) & (kOwn - 1)], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    a.x[row] = __longlong_as_double((long long)out);
                    __hip_atomic_store(&a.xs[slot0 + slot_b + wl], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else
                if (lane < w) {
                    __hip_atomic_store(&opnd[(slot_b + lane) & (kOwn - 1)], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (!(exp_flags & 1)) a.x[row] = __longlong_as_double((long long)out);
                    if (!(exp_flags & 2)) __hip_atomic_store(&a.xs[slot0 + slot_b + lane], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (DBG) a.dbg_pub[slot0 + slot_b + lane] = (long long)__builtin_amdgcn_s_memrealtime();
                }
                stamp(s, 6);
                if (__builtin_expect((s & 7) == 7, 0) && lane == 0) {
                    lds_release(&ctl[C_Q_DONE], (unsigned)quad_b);
                    lds_release(&ctl[C_SLOT_DONE], (unsigned)slot_b);
                    lds_release(&ctl[C_EXT_SAFE], (unsigned)(ext_prev - kExtWindow));
                }
                ext_prev = ext_end;
            }
            };
            // (the first batch of descriptors tells; a tile of more than 64 steps takes the general loop)
            const int nq_mine = lane < n_steps ? (d_cur.y >> 8) : -1;
            const int nq_hi = -wave_min_int(-nq_mine), nq_lo = wave_min_int(lane < n_steps ? nq_mine : 0x7fffffff);
            const int nq_all = __builtin_amdgcn_readfirstlane((n_steps <= 64 && nq_lo == nq_hi) ? nq_hi : 0);
            if (nq_all == 1) run_steps(std::integral_constant<int, 1>{});
            else if (nq_all == 2) run_steps(std::integral_constant<int, 2>{});
            else if (nq_all == 4 && CFG::kGroup == 4) run_steps(std::integral_constant<int, 4>{});
            else run_steps(std::integral_constant<int, 0>{});
            if (DBG && lane == 0) {
                long long *d = a.dbg + (int64_t)t * 16;
                d[0] = t_start; d[1] = (long long)__builtin_amdgcn_s_memrealtime(); d[2] = w_load; d[3] = w_ext;
                d[7] = (long long)n_steps | (((long long)__builtin_readcyclecounter() - c_first) << 16); // steps | core cycles from the first step to the last
            }
        } else if (wave == 1) {
            // ---- quad loader: the tile's entry stream (step order, quad-major / lane-minor) into the ring; the next
            // round's loads are in flight while this round waits for ring space and is written ----
            constexpr int U = kQuadChunk / 64;
            int4 c_n[U];
            double2 va_n[U], vb_n[U];
            auto issue = [&](int done, int chunk) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t g = quad0 + done + min(u * 64 + lane, chunk - 1);
                    // (read once: non-temporal, so the stream does not push the x / b / D lines out of the L2)
                    c_n[u] = nt_load(&a.quad_code[g]);
                    va_n[u] = nt_load(&a.quad_val[2 * g]);
                    vb_n[u] = nt_load(&a.quad_val[2 * g + 1]);
                }
            };
            if (exp_flags & 8) {
#pragma unroll
                for (int u = 0; u < U; ++u) { c_n[u] = make_int4(8 * kZeroSlot, 8 * kZeroSlot, 8 * kZeroSlot, 8 * kZeroSlot); va_n[u] = vb_n[u] = make_double2(0.0, 0.0); }
            }
            if (n_quads > 0 && !(exp_flags & 8)) issue(0, min(kQuadChunk, n_quads));
            for (int done = 0; done < n_quads;) {
                const int chunk = min(kQuadChunk, n_quads - done);
                int4 c[U];
                double2 va[U], vb[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { c[u] = c_n[u]; va[u] = va_n[u]; vb[u] = vb_n[u]; }
                if (done + chunk < n_quads && !(exp_flags & 8)) issue(done + chunk, min(kQuadChunk, n_quads - done - chunk));
                unsigned spins = 0;
                while (done + chunk - (int)lds_acquire(&ctl[C_Q_DONE]) > kRingQ) {
                    if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = u * 64 + lane;
                    if (i < chunk) {
                        const int r = (done + i) & (kRingQ - 1);
                        ring_q[r].c = c[u];
                        ring_q[r].va = va[u];
                        ring_q[r].vb = vb[u];
                    }
                }
                done += chunk;
                lds_release(&ctl[C_Q_LOADED], (unsigned)done);
            }
            if (DBG && lane == 0) a.dbg[(int64_t)t * 16 + 4] = (long long)__builtin_amdgcn_s_memrealtime();
        } else if (wave == 2) {
            // ---- per-row operand loader: row index, b[row], D[row] in slot order (rows one round ahead of their b / D) ----
            constexpr int U = kSlotChunk / 64;
            int r_next[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r_next[u] = a.slot_row[slot0 + min(u * 64 + lane, n_slots - 1)];
            for (int done = 0; done < n_slots;) {
                const int chunk = min(kSlotChunk, n_slots - done);
                int r[U];
                double bv[U], dv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    r[u] = r_next[u];
                    bv[u] = (exp_flags & 4) ? 1.0 : a.b[r[u]];
                    dv[u] = (exp_flags & 4) ? 2.0 : a.D[r[u]];
                }
                if (done + chunk < n_slots) {
#pragma unroll
                    for (int u = 0; u < U; ++u) r_next[u] = a.slot_row[slot0 + min(done + chunk + u * 64 + lane, n_slots - 1)];
                }
                unsigned spins = 0;
                while (done + chunk - (int)lds_acquire(&ctl[C_SLOT_DONE]) > kRingSlot) {
                    if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = u * 64 + lane;
                    if (i < chunk) {
                        ring_row[(done + i) & (kRingSlot - 1)] = r[u];
                        ring_bD[(done + i) & (kRingSlot - 1)] = make_double2(bv[u], dv[u]);
                    }
                }
                done += chunk;
                lds_release(&ctl[C_SLOT_LOADED], (unsigned)done);
            }
            if (DBG && lane == 0) a.dbg[(int64_t)t * 16 + 5] = (long long)__builtin_amdgcn_s_memrealtime();
        } else {
            // ---- poller: external operands in first-need order into their ring, a block of kPollBlock ordinals (8 per
            // lane, all in flight together) at a time; the watermark (all ordinals below it delivered) moves when a
            // block is complete ----
            constexpr int U = kPollBlock / 64;
            int wm_pub = 0, idle = 0;
            long long rounds = 0;
            const long long c_poll = DBG ? (long long)__builtin_readcyclecounter() : 0;
            // (the producing slots of the NEXT block are fetched while this block is polled: a block boundary otherwise costs a
            // trip to memory during which nothing is looked for)
            int src_next[U];
#pragma unroll
            for (int u = 0; u < U; ++u) src_next[u] = u * 64 + lane < n_ext ? a.ext_src[ext0 + u * 64 + lane] : 0;
            for (int base = 0; base < n_ext; base += kPollBlock) {
                int src[U];
                bool got[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = base + u * 64 + lane;
                    got[u] = e >= n_ext;
                    src[u] = src_next[u];
                    const int e_next = e + kPollBlock;
                    src_next[u] = e_next < n_ext ? a.ext_src[ext0 + e_next] : 0;
                }
                // the ring positions of this block are free once the ordinals kExt before them are dead
                unsigned spins = 0;
                while (min(base + kPollBlock, n_ext) > (int)lds_acquire(&ctl[C_EXT_SAFE]) + kExt) {
                    if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                spins = 0;
                for (;;) {
                    unsigned long long vb[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) vb[u] = got[u] ? 0ull : __hip_atomic_load(&a.xs[src[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // ... or as soon as any wait of the sweep has raised the fault word: the grid then drains at once
                    ++spins;
                    const bool give_up = (exp_flags & 16) || spins > kSpinMem || ((spins & 255u) == 0u && __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u);
                    bool all = true;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (!got[u] && (vb[u] != kSentinel || give_up)) {
                            __hip_atomic_store(&opnd[kOwn + ((base + u * 64 + lane) & (kExt - 1))], vb[u] == kSentinel ? kCanonNaN : vb[u],
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            got[u] = true;
                            if (DBG) a.dbg_del[ext0 + base + u * 64 + lane] = (long long)__builtin_amdgcn_s_memrealtime();
                        }
                        all &= got[u];
                    }
                    if (give_up && !(exp_flags & 16) && lane == 0) __hip_atomic_fetch_or(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    // watermark = the first ordinal of the block not delivered yet (ordinals run lane-minor): a step may need
                    // an early ordinal of this block to produce a later one of the same block
                    int wm = min(base + kPollBlock, n_ext);
#pragma unroll
                    for (int u = U - 1; u >= 0; --u) {
                        const unsigned long long open = __ballot(!got[u]);
                        if (open) wm = base + u * 64 + (int)__builtin_ctzll(open);
                    }
                    const bool moved = wm != wm_pub;
                    if (moved) { wm_pub = wm; if (lane == 0) lds_release(&ctl[C_EXT_WM], (unsigned)wm); }
                    if (DBG) ++rounds;
                    if (!__ballot(!all)) break;
                    __builtin_amdgcn_s_sleep(1);
                    // a poller whose rounds deliver nothing backs off (up to a.backoff x 64 cycles between rounds): most resident tiles
                    // are far ahead of the wavefront, and their polling is load in the L2 the active tiles' rounds queue behind
                    idle = moved ? 0 : min(idle + 1, a.backoff);
                    for (int k = 0; k < idle; ++k) __builtin_amdgcn_s_sleep(1);
                }
            }
            if (DBG && lane == 0) {
                a.dbg[(int64_t)t * 16 + 6] = (long long)__builtin_amdgcn_s_memrealtime();
                a.dbg[(int64_t)t * 16 + 8] = rounds;
                a.dbg[(int64_t)t * 16 + 9] = (long long)__builtin_readcyclecounter() - c_poll;
            }
        }
        __syncthreads();
    }
}

} // namespace

// ---- plan (host analysis) ---------------------------------------------------------------------------
// Input: the strictly triangular pattern on the host.  Output: the processing order, its tiles, and per tile the
// steps, the repacked entries and the list of external operands.
static bis_status trsv_tiled_build_host(bis_ctx *ctx, const bis_mat *T, bool backward, bis_trsv_tiled **out) {
    *out = nullptr;
    const int64_t n = T->n_rows;
    if (n == 0 || T->nnz == 0 || T->nnz > (int64_t)600000000 || T->view) return BIS_OK; // not applicable: caller keeps the level-scheduled sweep
    std::vector<int64_t> rp(n + 1);
    std::vector<int32_t> col((size_t)T->nnz);
    bis_status st = bis_mat_download(ctx, T, rp.data(), col.data(), nullptr);
    if (st != BIS_OK) return st;
    int max_len = 0;
    for (int64_t r = 0; r < n; ++r) max_len = std::max<int>(max_len, (int)(rp[r + 1] - rp[r]));
    const int kOwn = Cfg::kOwn, kExt = Cfg::kExt, kExtWindow = Cfg::kExtWindow, kRingQ = Cfg::kRingQ;
    const int kZeroSlot = kOwn + kExt;
    if ((max_len + 3) / 4 > kRingQ / 2) return BIS_OK; // a single row must fit half the quad ring
    // The processing order: ord[pos] = row, any linear extension of the dependency order (every operand of a row sits
    // at an earlier position).  Default: the reference's substitution order (ascending rows forward, descending
    // backward) cut into intervals.  With a grid hint: tiles that extend in ALL grid directions.
    std::vector<int32_t> ord((size_t)n), pos_v((size_t)n);
    std::vector<int64_t> tile_pos0;
    bool grid_tiles = false;
    // tile extents in nodes: trsv_tile_edge = e (cubic) or ex | ey << 8 | ez << 16.  Defaults from tools/trsv_ab.py with the
    // 512-quad ring, forward / backward sweep in ms:
    //   7-point 256^3 (3 operands per row): 8x8x8 0.65 / 0.65; 16x8x8 0.74; 8x16x8 0.79; 8x8x16 0.75; 16x16x8 1.13; 8x8x4 1.69
    //   27-point 256^3 (13):                8x8x4 1.75 / 1.78; 8x4x4 2.14; 8x4x8 1.88; 16x8x4 1.92; 8x16x4 2.03; 8^3 2.12; 8x8x2 2.19
    //   FEM-like 80x80x81x3 (~35):          2x2x2 2.28 / 2.28; 3^3 2.55
    // a hop costs the tile's extent in levels plus a memory round trip; a tile costs its start-up (ticket, descriptors, first
    // loader rounds: ~4 us); the deeper the quad ring, the larger the tile that still runs at the compute wave's pace
    const int edge_default = max_len <= 8 ? (8 | 8 << 8 | 8 << 16) : max_len <= 16 ? (8 | 8 << 8 | 4 << 16) : (2 | 2 << 8 | 2 << 16);
    const int edge_opt = bis_opts().trsv_tile_edge >= 0 ? bis_opts().trsv_tile_edge : edge_default;
    const int ex = edge_opt < 256 ? edge_opt : (edge_opt & 255), ey = edge_opt < 256 ? edge_opt : ((edge_opt >> 8) & 255), ez = edge_opt < 256 ? edge_opt : ((edge_opt >> 16) & 255);
    const int edge = std::min(ex, std::min(ey, ez));
    if (T->grid[0] > 0 && edge > 0 && T->grid[0] * T->grid[1] * T->grid[2] * T->grid[3] == n) {
        // Skewed tile-major order.  Node coordinates are mirrored for the backward sweep, so that in both directions an
        // operand has a SMALLER node (or the same node and a smaller unknown).  With x' = x + a y + b z, y' = y + c z,
        // z' = z and the smallest skews a, b, c for which every operand offset has x' <= 0, y' <= 0, z' <= 0, the order
        // (tile of (z', y', x'), then (z', y', x', unknown) inside the tile) is a linear extension: 7-point stencils need
        // no skew, 27-point ones a = 1, b = 2, c = 1.  The order is verified below; if it is not one (a pattern the
        // hint does not describe), the interval tiles of the natural order are used.
        const int64_t nx = T->grid[0], ny = T->grid[1], nz = T->grid[2], dof = T->grid[3];
        auto node_of = [&](int64_t row, int64_t &x, int64_t &y, int64_t &z, int64_t &d) {
            d = row % dof;
            const int64_t a0 = row / dof;
            x = a0 % nx; y = (a0 / nx) % ny; z = a0 / (nx * ny);
            if (backward) { x = nx - 1 - x; y = ny - 1 - y; z = nz - 1 - z; d = dof - 1 - d; }
        };
        // distinct operand offsets (sampled on every row: O(nnz), cheap next to the rest of the plan)
        std::vector<int64_t> offs; // packed (dx, dy, dz)
        {
            std::vector<int64_t> seen;
            for (int64_t r = 0; r < n; ++r) {
                int64_t x, y, z, d;
                node_of(r, x, y, z, d);
                for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
                    int64_t cx, cy, cz, cd;
                    node_of(col[k], cx, cy, cz, cd);
                    const int64_t key = ((cz - z + nz) * (2 * ny + 1) + (cy - y + ny)) * (2 * nx + 1) + (cx - x + nx);
                    if (seen.size() < 128 && std::find(seen.begin(), seen.end(), key) == seen.end()) seen.push_back(key);
                }
                if (seen.size() >= 128) break; // not a stencil: no grid tiles
            }
            offs.swap(seen);
        }
        int sa = -1, sb = -1, sc = -1;
        if (offs.size() < 128) {
            for (int tot = 0; tot <= 9 && sa < 0; ++tot)
                for (int a2 = 0; a2 <= 3 && sa < 0; ++a2)
                    for (int b2 = 0; b2 <= 3 && sa < 0; ++b2) {
                        const int c2 = tot - a2 - b2;
                        if (c2 < 0 || c2 > 3) continue;
                        bool ok = true;
                        for (int64_t key : offs) {
                            const int64_t dx = key % (2 * nx + 1) - nx, dy = (key / (2 * nx + 1)) % (2 * ny + 1) - ny, dz = key / ((2 * nx + 1) * (2 * ny + 1)) - nz;
                            if (dz > 0 || dy + c2 * dz > 0 || dx + a2 * dy + b2 * dz > 0) { ok = false; break; }
                        }
                        if (ok) { sa = a2; sb = b2; sc = c2; }
                    }
        }
        if (sa >= 0) {
            const int64_t NXs = (nx + sa * (ny - 1) + sb * (nz - 1) + ex - 1) / ex, NYs = (ny + sc * (nz - 1) + ey - 1) / ey;
            const uint64_t n_lex = (uint64_t)(((nz + ez - 1) / ez) * NYs * NXs);
            std::vector<std::pair<uint64_t, int32_t>> keyed((size_t)n);
            for (int64_t r = 0; r < n; ++r) {
                int64_t x, y, z, d;
                node_of(r, x, y, z, d);
                const int64_t xs = x + sa * y + sb * z, ys = y + sc * z, zs = z;
                // tiles are numbered hyperplane by hyperplane (X' + Y' + Z' = const), not slab by slab: the tickets hand them
                // out in this order, and the tiles a persistent grid holds at a time must be the ones that can run together
                // (slab-major numbering left ~30 of 512 resident tiles runnable: 5 ms per sweep instead of 0.6)
                const uint64_t lex = (uint64_t)(((zs / ez) * NYs + ys / ey) * NXs + xs / ex);
                const uint64_t tile = (uint64_t)(zs / ez + ys / ey + xs / ex) * n_lex + lex;
                const uint64_t intra = (uint64_t)((((zs % ez) * ey + ys % ey) * ex + xs % ex) * dof + d);
                keyed[(size_t)r] = {tile * (uint64_t)((int64_t)ex * ey * ez * dof) + intra, (int32_t)r};
            }
            std::sort(keyed.begin(), keyed.end());
            const uint64_t tile_vol = (uint64_t)((int64_t)ex * ey * ez * dof);
            bool fits = tile_vol <= (uint64_t)kMaxB;
            for (int64_t p = 0; p < n; ++p) { ord[(size_t)p] = keyed[(size_t)p].second; pos_v[(size_t)keyed[(size_t)p].second] = (int32_t)p; }
            // a linear extension?
            bool valid = fits;
            for (int64_t r = 0; r < n && valid; ++r)
                for (int64_t k = rp[r]; k < rp[r + 1]; ++k)
                    if (pos_v[(size_t)col[k]] >= pos_v[(size_t)r]) { valid = false; break; }
            if (valid) {
                tile_pos0.push_back(0);
                for (int64_t p = 1; p < n; ++p)
                    if (keyed[(size_t)p].first / tile_vol != keyed[(size_t)p - 1].first / tile_vol) tile_pos0.push_back(p);
                tile_pos0.push_back(n);
                grid_tiles = true;
            }
        }
    }
    if (!grid_tiles) {
        for (int64_t p = 0; p < n; ++p) { const int64_t r = backward ? n - 1 - p : p; ord[(size_t)p] = (int32_t)r; pos_v[(size_t)r] = (int32_t)p; }
    }
    auto row_at = [&](int64_t pos) { return (int64_t)ord[(size_t)pos]; };
    auto pos_of = [&](int64_t row) { return (int64_t)pos_v[(size_t)row]; };
    // measured (tools/trsv_ab.py): 7-point 256^3 1.46 ms at 8192 (1.65 at 4096, 1.83 at 16384); 27-point 128^3 1.18 ms at 1024, 1.24 at 2048, 1.86 at 8192
    const int max_rows = bis_opts().trsv_tile_rows > 0 ? std::min(bis_opts().trsv_tile_rows, kMaxB) : (max_len <= 8 ? 8192 : 2048);
    // pass A (interval tiles of the natural order): tile boundaries.  Within the second half of its allowed extent a
    // tile is cut where the next tile's first row depends only on results its predecessor produces EARLY (small local
    // level).  Cutting in the middle of a grid line would make every tile wait for the end of its predecessor and
    // serialise the sweep (measured: 494 ms instead of 4 ms on the 256^3 7-point grid); cutting at the start of a
    // grid line lets it start on the predecessor's early results.
    if (!grid_tiles) {
        tile_pos0.push_back(0);
        std::vector<int> lv((size_t)max_rows + 1);
        for (int64_t p0 = 0; p0 < n;) {
            const int64_t p = std::min<int64_t>(n, p0 + max_rows);
            int64_t cut = p;
            if (p < n) {
                const int m = (int)(p - p0);
                for (int i = 0; i <= m; ++i) { // i == m: row p itself (the cut that keeps the whole extent)
                    const int64_t r = row_at(p0 + i);
                    int l = 0;
                    for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
                        const int64_t q = pos_of(col[k]);
                        if (q >= p0) l = std::max(l, lv[(size_t)(q - p0)] + 1);
                    }
                    lv[(size_t)i] = l;
                }
                const int half = std::max(1, m / 2);
                int best = INT32_MAX;
                for (int c = half; c <= m; ++c) best = std::min(best, lv[(size_t)c]);
                const int accept = best + best / 2 + 8; // the largest tile whose successor starts about as early as the best cut allows
                for (int c = m; c >= half; --c)
                    if (lv[(size_t)c] <= accept) { cut = p0 + c; break; }
            }
            tile_pos0.push_back(cut);
            p0 = cut;
        }
    }
    const int64_t n_tiles = (int64_t)tile_pos0.size() - 1;
    if (n_tiles > INT32_MAX) return BIS_OK;
    std::vector<int32_t> tile_of((size_t)n);
    for (int64_t t = 0; t < n_tiles; ++t)
        for (int64_t p = tile_pos0[t]; p < tile_pos0[t + 1]; ++p) tile_of[(size_t)p] = (int32_t)t;
    // pass B: local levels, steps, quads, external ordinals
    std::vector<int32_t> slot_row((size_t)n);
    std::vector<int4> step_desc;
    std::vector<int64_t> tile_step0(n_tiles + 1, 0), tile_quad0(n_tiles + 1, 0), tile_ext0(n_tiles + 1, 0);
    std::vector<int4> quad_code;
    std::vector<int64_t> quad_src; // CRS index of each entry (4 per quad), -1 = padding
    std::vector<int32_t> ext_src;
    std::vector<int32_t> lidx((size_t)n); // row -> slot within its tile
    step_desc.reserve((size_t)(n / 16));
    quad_code.reserve((size_t)(T->nnz / 3));
    quad_src.reserve((size_t)(T->nnz / 3) * 4);
    std::vector<int> lvl(kMaxB), order(kMaxB), cnt;
    std::vector<int32_t> ext_stamp((size_t)n, -1), ext_ord((size_t)n, 0); // per global slot: tile that listed it last, its ordinal there
    int tile_rows_max = 0;
    int64_t n_demoted = 0;
    for (int64_t t = 0; t < n_tiles; ++t) {
        const int64_t p0 = tile_pos0[t];
        const int m = (int)(tile_pos0[t + 1] - p0);
        tile_rows_max = std::max(tile_rows_max, m);
        int max_lvl = 0;
        for (int i = 0; i < m; ++i) { // in-tile dependencies sit at earlier positions of the tile
            const int64_t r = row_at(p0 + i);
            int l = 0;
            for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
                const int64_t q = pos_of(col[k]);
                if (q >= p0) l = std::max(l, lvl[(int)(q - p0)] + 1);
            }
            lvl[i] = l;
            max_lvl = std::max(max_lvl, l);
        }
        cnt.assign(max_lvl + 2, 0); // stable counting sort by level
        for (int i = 0; i < m; ++i) cnt[lvl[i] + 1]++;
        for (int l = 0; l <= max_lvl; ++l) cnt[l + 1] += cnt[l];
        {
            std::vector<int> fill(cnt.begin(), cnt.end() - 1);
            for (int i = 0; i < m; ++i) order[fill[lvl[i]]++] = i;
        }
        for (int s = 0; s < m; ++s) {
            const int64_t r = row_at(p0 + order[s]);
            slot_row[(size_t)(p0 + s)] = (int32_t)r;
            lidx[(size_t)r] = s;
        }
        tile_step0[t] = (int64_t)step_desc.size();
        tile_quad0[t] = (int64_t)quad_code.size();
        tile_ext0[t] = (int64_t)ext_src.size();
        int n_ext_tile = 0;
        for (int l = 0; l <= max_lvl; ++l) {
            int s = cnt[l];
            const int s_end = cnt[l + 1];
            while (s < s_end) { // steps: runs of one level, at most 64 rows, at most kRingQ/2 quads
                int w = 0, nq = 0;
                while (s + w < s_end && w < 64) {
                    const int64_t r = slot_row[(size_t)(p0 + s + w)];
                    const int nq2 = std::max(nq, (int)((rp[r + 1] - rp[r] + 3) / 4));
                    if (w > 0 && (w + 1) * nq2 > kRingQ / 2) break;
                    nq = nq2;
                    ++w;
                }
                const int quad_b = (int)((int64_t)quad_code.size() - tile_quad0[t]);
                const int ext_base = n_ext_tile; // ordinals first needed before this step
                for (int g = 0; g < nq; ++g)
                    for (int i = 0; i < w; ++i) {
                        const int64_t r = slot_row[(size_t)(p0 + s + i)];
                        int code[4];
                        for (int q = 0; q < 4; ++q) {
                            const int64_t kk = rp[r] + 4 * g + q;
                            if (kk >= rp[r + 1]) { code[q] = kZeroSlot; quad_src.push_back(-1); continue; }
                            quad_src.push_back(kk);
                            const int64_t qp = pos_of(col[kk]);
                            const int ps = lidx[(size_t)col[kk]]; // the operand's slot within ITS tile
                            if (qp >= p0 && (s + i) - ps <= kOwn - 128) {
                                code[q] = ps & (kOwn - 1); // still in the ring of the tile's own results
                            } else { // another tile's result, or one of this tile's that has left the ring: through the poller
                                if (qp >= p0) ++n_demoted;
                                const int32_t gs = (int32_t)(tile_pos0[tile_of[(size_t)qp]] + ps);
                                if (ext_stamp[(size_t)gs] != (int32_t)t || ext_ord[(size_t)gs] < ext_base - kExtWindow) {
                                    ext_stamp[(size_t)gs] = (int32_t)t; // first need, or last listed too long ago: next ordinal
                                    ext_ord[(size_t)gs] = n_ext_tile++;
                                    ext_src.push_back(gs);
                                }
                                code[q] = kOwn + (ext_ord[(size_t)gs] & (kExt - 1));
                            }
                        }
                        quad_code.push_back(make_int4(8 * code[0], 8 * code[1], 8 * code[2], 8 * code[3])); // byte offsets into the operand array
                    }
                step_desc.push_back(make_int4(s, w | (nq << 8), quad_b, n_ext_tile));
                s += w;
            }
        }
    }
    tile_step0[n_tiles] = (int64_t)step_desc.size();
    tile_quad0[n_tiles] = (int64_t)quad_code.size();
    tile_ext0[n_tiles] = (int64_t)ext_src.size();
    // upload
    bis_trsv_tiled *p = new bis_trsv_tiled;
    p->n = n; p->n_tiles = (int)n_tiles; p->max_rows = tile_rows_max;
    p->n_steps = (int64_t)step_desc.size();
    p->n_quads = (int64_t)quad_code.size();
    p->n_ext = (int64_t)ext_src.size();
    int64_t *d_src = nullptr;
    hipError_t e = hipSuccess;
    auto up = [&](void **dst, const void *src, size_t bytes) {
        if (e != hipSuccess) return;
        e = hipMalloc(dst, std::max<size_t>(bytes, 16));
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    };
    up((void **)&p->slot_row, slot_row.data(), 4 * slot_row.size());
    up((void **)&p->step_desc, step_desc.data(), sizeof(int4) * step_desc.size());
    up((void **)&p->tile_slot0, tile_pos0.data(), 8 * tile_pos0.size());
    up((void **)&p->tile_step0, tile_step0.data(), 8 * tile_step0.size());
    up((void **)&p->tile_quad0, tile_quad0.data(), 8 * tile_quad0.size());
    up((void **)&p->tile_ext0, tile_ext0.data(), 8 * tile_ext0.size());
    up((void **)&p->quad_code, quad_code.data(), sizeof(int4) * quad_code.size());
    up((void **)&p->ext_src, ext_src.data(), 4 * ext_src.size());
    up((void **)&d_src, quad_src.data(), 8 * quad_src.size());
    if (e == hipSuccess) e = hipMalloc(&p->quad_val, sizeof(double) * std::max<size_t>(quad_src.size(), 4));
    if (e == hipSuccess) e = hipMalloc(&p->xs, sizeof(double) * (size_t)(n + 1));
    if (e == hipSuccess) e = hipMalloc(&p->ticket, sizeof(unsigned) * 4);
    if (e == hipSuccess && !quad_src.empty()) {
        hipLaunchKernelGGL(gather_entries_kernel, dim3((unsigned)std::min<int64_t>(((int64_t)quad_src.size() + 255) / 256, 8192)), dim3(256), 0,
                           ctx->stream, T->val, d_src, (int64_t)quad_src.size(), (double *)p->quad_val);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d_src);
    if (e != hipSuccess) {
        ctx->err = std::string("tiled sptrsv plan: ") + hipGetErrorString(e);
        bis_trsv_tiled_destroy(p);
        return BIS_ERR_HIP;
    }
    if (getenv("BIS_TRSV_TILE_STATS"))
        fprintf(stderr, "tiled sptrsv plan (%s): %lld rows, %d tiles (largest %d rows), %lld steps (%.1f rows per step), %lld quads (%.2f x the entries), "
                        "%lld external ordinals (%.2f per row), %lld in-tile operands beyond the ring%s\n", backward ? "backward" : "forward",
                (long long)n, p->n_tiles, tile_rows_max, (long long)p->n_steps, (double)n / (double)std::max<int64_t>(p->n_steps, 1),
                (long long)p->n_quads, 4.0 * (double)p->n_quads / (double)T->nnz, (long long)p->n_ext, (double)p->n_ext / (double)n,
                (long long)n_demoted, grid_tiles ? "; grid tiles" : "; interval tiles");
    *out = p;
    return BIS_OK;
}

// ---- plan (device analysis, grid-hinted matrices) ----------------------------------------------------
// The same plan as trsv_tiled_build_host, array for array, built where the matrix lives: the skew search is a bit
// mask reduced over all entries, the order a radix sort of the tile keys, and the per-tile work (local levels, steps,
// quads, external ordinals) the host's sequential loop run by ONE LANE PER TILE -- tens of thousands of tiles run
// side by side, which is all the parallelism this needs (measured: DESIGN.md section 4).  Sizes are data dependent, so
// the per-tile pass runs twice: count, scan, fill.
namespace {

struct PlanGeom {
    long long nx, ny, nz, dof, NXs, NYs;
    unsigned long long n_lex, tile_vol;
    int backward, sa, sb, sc, ex, ey, ez;
};

__device__ __forceinline__ void plan_node(const PlanGeom &g, long long row, long long &x, long long &y, long long &z, long long &d) {
    d = row % g.dof;
    const long long a0 = row / g.dof;
    x = a0 % g.nx; y = (a0 / g.nx) % g.ny; z = a0 / (g.nx * g.ny);
    if (g.backward) { x = g.nx - 1 - x; y = g.ny - 1 - y; z = g.nz - 1 - z; d = g.dof - 1 - d; }
}

// bit (a + 4 b + 16 c) of *mask survives iff every operand offset has x' <= 0, y' <= 0, z' <= 0 under the skew (a, b, c)
template <class RP>
__global__ __launch_bounds__(256) void plan_skew_kernel(PlanGeom g, int64_t n, const RP *__restrict__ rp, const int32_t *__restrict__ col,
                                                        unsigned long long *mask) {
    unsigned long long m = ~0ull;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
        long long x, y, z, d;
        plan_node(g, r, x, y, z, d);
        for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
            long long cx, cy, cz, cd;
            plan_node(g, col[k], cx, cy, cz, cd);
            const long long dx = cx - x, dy = cy - y, dz = cz - z;
            if (dz > 0) { m = 0; continue; }
            for (int c = 0; c < 4; ++c) {
                if (dy + c * dz > 0) { m &= ~(0xFFFFull << (16 * c)); continue; }
                for (int b = 0; b < 4; ++b)
                    for (int a = 0; a < 4; ++a)
                        if (dx + a * dy + b * dz > 0) m &= ~(1ull << (a + 4 * b + 16 * c));
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) m &= __shfl_xor(m, o);
    if ((threadIdx.x & 63) == 0 && m != ~0ull) atomicAnd(mask, m);
}

__global__ __launch_bounds__(256) void plan_key_kernel(PlanGeom g, int64_t n, unsigned long long *__restrict__ key, int32_t *__restrict__ row) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
        long long x, y, z, d;
        plan_node(g, r, x, y, z, d);
        const long long xs = x + g.sa * y + g.sb * z, ys = y + g.sc * z, zs = z;
        const unsigned long long lex = (unsigned long long)(((zs / g.ez) * g.NYs + ys / g.ey) * g.NXs + xs / g.ex);
        const unsigned long long tile = (unsigned long long)(zs / g.ez + ys / g.ey + xs / g.ex) * g.n_lex + lex;
        const unsigned long long intra = (unsigned long long)((((zs % g.ez) * g.ey + ys % g.ey) * g.ex + xs % g.ex) * g.dof + d);
        key[r] = tile * g.tile_vol + intra;
        row[r] = (int32_t)r;
    }
}

template <class RP>
__global__ __launch_bounds__(256) void plan_pos_kernel(int64_t n, unsigned long long tile_vol, const unsigned long long *__restrict__ key,
                                                       const int32_t *__restrict__ ord, const RP *__restrict__ rp, int32_t *__restrict__ pos,
                                                       int32_t *__restrict__ flag, int64_t *__restrict__ len) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += stride) {
        const int32_t r = ord[p];
        pos[r] = (int32_t)p;
        flag[p] = (p == 0 || key[p] / tile_vol != key[p - 1] / tile_vol) ? 1 : 0;
        len[p] = (int64_t)(rp[r + 1] - rp[r]);
    }
}

__global__ __launch_bounds__(256) void plan_tiles_kernel(int64_t n, int64_t n_tiles, const int32_t *__restrict__ flag, const int32_t *__restrict__ tix,
                                                         int64_t *__restrict__ tile_pos0) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += stride) {
        if (flag[p]) tile_pos0[tix[p] - 1] = p;
        if (p == 0) tile_pos0[n_tiles] = n;
    }
}

// the order must be a linear extension of the dependency order
template <class RP>
__global__ __launch_bounds__(256) void plan_verify_kernel(int64_t n, const RP *__restrict__ rp, const int32_t *__restrict__ col,
                                                          const int32_t *__restrict__ pos, int *bad) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    bool b = false;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
        const int32_t pr = pos[r];
        for (int64_t k = rp[r]; k < rp[r + 1]; ++k) b |= pos[col[k]] >= pr;
    }
    if (b) *bad = 1;
}

struct PlanArgs {
    int64_t n, nnz, n_tiles;
    const void *rp;
    const int32_t *col;
    const double *val;
    const int64_t *tile_pos0;
    const int32_t *ord, *pos, *tix;
    const int64_t *entpos;
    int32_t *lvl, *cnt, *fill, *order, *lidx, *tile_maxlvl;
    int32_t *slot_row;
    int32_t *hash_key, *hash_ord;
    int64_t *n_steps, *n_quads, *n_ext; // per tile: counts (pass 1), then their exclusive scans (pass 2)
    int4 *step_desc, *quad_code;
    double *quad_val;
    int32_t *ext_src;
    unsigned long long *n_demoted;
    int kOwn, kExt, kExtWindow, kRingQ, kZeroSlot;
};

// local levels and the (level, processing order) sort of each tile: slot_row, lidx
template <class RP>
__global__ __launch_bounds__(64) void plan_levels_kernel(PlanArgs a) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= a.n_tiles) return;
    const RP *rp = (const RP *)a.rp;
    const int64_t p0 = a.tile_pos0[t];
    const int m = (int)(a.tile_pos0[t + 1] - p0);
    int32_t *lvl = a.lvl + p0, *order = a.order + p0, *cnt = a.cnt + p0 + 2 * t, *fill = a.fill + p0 + 2 * t;
    int max_lvl = 0;
    for (int i = 0; i < m; ++i) { // in-tile dependencies sit at earlier positions of the tile
        const int64_t r = a.ord[p0 + i];
        int l = 0;
        for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
            const int64_t q = a.pos[a.col[k]];
            if (q >= p0) l = max(l, lvl[(int)(q - p0)] + 1);
        }
        lvl[i] = l;
        max_lvl = max(max_lvl, l);
    }
    for (int l = 0; l < max_lvl + 2; ++l) cnt[l] = 0; // stable counting sort by level
    for (int i = 0; i < m; ++i) cnt[lvl[i] + 1]++;
    for (int l = 0; l <= max_lvl; ++l) cnt[l + 1] += cnt[l];
    for (int l = 0; l <= max_lvl; ++l) fill[l] = cnt[l];
    for (int i = 0; i < m; ++i) order[fill[lvl[i]]++] = i;
    for (int s = 0; s < m; ++s) {
        const int32_t r = a.ord[p0 + order[s]];
        a.slot_row[p0 + s] = r;
        a.lidx[r] = s;
    }
    a.tile_maxlvl[t] = max_lvl;
}

// steps, quads and external ordinals of each tile; FILL = false counts, FILL = true writes at the scanned offsets
template <class RP, bool FILL>
__global__ __launch_bounds__(64) void plan_steps_kernel(PlanArgs a) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= a.n_tiles) return;
    const RP *rp = (const RP *)a.rp;
    const int64_t p0 = a.tile_pos0[t];
    const int m = (int)(a.tile_pos0[t + 1] - p0);
    const int32_t *cnt = a.cnt + p0 + 2 * t;
    const int max_lvl = a.tile_maxlvl[t];
    const int64_t e0 = a.entpos[p0];
    const int64_t E = (p0 + m < a.n ? a.entpos[p0 + m] : a.nnz) - e0;
    unsigned cap = 4; // the tile's own hash table (slot of an external operand -> its ordinal): 4 E words reserved, at most half full
    while ((int64_t)cap < 2 * E) cap <<= 1;
    int32_t *hk = a.hash_key + 4 * e0, *ho = a.hash_ord + 4 * e0;
    const int64_t step0 = FILL ? a.n_steps[t] : 0, quad0 = FILL ? a.n_quads[t] : 0, ext0 = FILL ? a.n_ext[t] : 0;
    int64_t step_i = 0, quad_i = 0;
    int n_ext_tile = 0;
    unsigned long long demoted = 0;
    for (int l = 0; l <= max_lvl; ++l) {
        int s = cnt[l];
        const int s_end = cnt[l + 1];
        while (s < s_end) { // steps: runs of one level, at most 64 rows, at most kRingQ/2 quads
            int w = 0, nq = 0;
            while (s + w < s_end && w < 64) {
                const int64_t r = a.slot_row[p0 + s + w];
                const int nq2 = max(nq, (int)((rp[r + 1] - rp[r] + 3) / 4));
                if (w > 0 && (w + 1) * nq2 > a.kRingQ / 2) break;
                nq = nq2;
                ++w;
            }
            const int quad_b = (int)quad_i;
            const int ext_base = n_ext_tile; // ordinals first needed before this step
            for (int g = 0; g < nq; ++g)
                for (int i = 0; i < w; ++i) {
                    const int64_t r = a.slot_row[p0 + s + i];
                    const int64_t k0 = rp[r], k1 = rp[r + 1];
                    int code[4];
                    double v[4];
                    for (int q = 0; q < 4; ++q) {
                        const int64_t kk = k0 + 4 * g + q;
                        if (kk >= k1) { code[q] = a.kZeroSlot; v[q] = 0.0; continue; }
                        const int32_t c = a.col[kk];
                        if (FILL) v[q] = a.val[kk];
                        const int64_t qp = a.pos[c];
                        const int ps = a.lidx[c]; // the operand's slot within ITS tile
                        if (qp >= p0 && (s + i) - ps <= a.kOwn - 128) {
                            code[q] = ps & (a.kOwn - 1); // still in the ring of the tile's own results
                        } else { // another tile's result, or one of this tile's that has left the ring: through the poller
                            if (qp >= p0) ++demoted;
                            const int32_t gs = (int32_t)(a.tile_pos0[a.tix[qp] - 1] + ps);
                            unsigned h = ((unsigned)gs * 2654435761u) & (cap - 1);
                            while (hk[h] != -1 && hk[h] != gs) h = (h + 1) & (cap - 1);
                            if (hk[h] != gs || ho[h] < ext_base - a.kExtWindow) { // first need, or last listed too long ago: next ordinal
                                hk[h] = gs;
                                ho[h] = n_ext_tile;
                                if (FILL) a.ext_src[ext0 + n_ext_tile] = gs;
                                ++n_ext_tile;
                            }
                            code[q] = a.kOwn + (ho[h] & (a.kExt - 1));
                        }
                    }
                    if (FILL) {
                        a.quad_code[quad0 + quad_i] = make_int4(8 * code[0], 8 * code[1], 8 * code[2], 8 * code[3]); // byte offsets into the operand array
                        double2 *qv = (double2 *)(a.quad_val + 4 * (quad0 + quad_i));
                        qv[0] = make_double2(v[0], v[1]);
                        qv[1] = make_double2(v[2], v[3]);
                    }
                    ++quad_i;
                }
            if (FILL) a.step_desc[step0 + step_i] = make_int4(s, w | (nq << 8), quad_b, n_ext_tile);
            ++step_i;
            s += w;
        }
    }
    if (!FILL) {
        a.n_steps[t] = step_i; a.n_quads[t] = quad_i; a.n_ext[t] = n_ext_tile;
        if (demoted) atomicAdd(a.n_demoted, demoted);
    }
}

struct DevBufs { // scratch of the device plan, freed on every way out
    std::vector<void *> v;
    hipError_t e = hipSuccess;
    template <class T> T *get(size_t count) {
        void *p = nullptr;
        if (e == hipSuccess && bis_opts().trsv_inject_oom > 0) e = hipErrorOutOfMemory; // test hook: the plan's scratch does not fit
        if (e == hipSuccess) e = hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16));
        if (e == hipSuccess) v.push_back(p); else p = nullptr;
        return (T *)p;
    }
    ~DevBufs() { for (void *p : v) hipFree(p); }
};

} // namespace

// *out stays null (BIS_OK) when the grid hint does not give a valid tile order
static bis_status trsv_tiled_build_device(bis_ctx *ctx, const bis_mat *T, bool backward, bis_trsv_tiled **out) {
    *out = nullptr;
    const int64_t n = T->n_rows;
    if (n == 0 || T->nnz == 0 || T->nnz > (int64_t)600000000 || T->view || n >= INT32_MAX) return BIS_OK;
    if (!(T->grid[0] > 0 && T->grid[0] * T->grid[1] * T->grid[2] * T->grid[3] == n)) return BIS_OK;
    const int max_len = T->max_row_nnz;
    PlanArgs a{};
    a.kOwn = Cfg::kOwn; a.kExt = Cfg::kExt; a.kExtWindow = Cfg::kExtWindow; a.kRingQ = Cfg::kRingQ;
    a.kZeroSlot = a.kOwn + a.kExt;
    if ((max_len + 3) / 4 > a.kRingQ / 2) return BIS_OK;
    const int edge_default = max_len <= 8 ? (8 | 8 << 8 | 8 << 16) : max_len <= 16 ? (8 | 8 << 8 | 4 << 16) : (2 | 2 << 8 | 2 << 16); // measured: trsv_tiled_build_host
    const int edge_opt = bis_opts().trsv_tile_edge >= 0 ? bis_opts().trsv_tile_edge : edge_default;
    PlanGeom g{};
    g.nx = T->grid[0]; g.ny = T->grid[1]; g.nz = T->grid[2]; g.dof = T->grid[3]; g.backward = backward ? 1 : 0;
    g.ex = edge_opt < 256 ? edge_opt : (edge_opt & 255); g.ey = edge_opt < 256 ? edge_opt : ((edge_opt >> 8) & 255);
    g.ez = edge_opt < 256 ? edge_opt : ((edge_opt >> 16) & 255);
    if (std::min(g.ex, std::min(g.ey, g.ez)) <= 0) return BIS_OK;
    g.tile_vol = (unsigned long long)((long long)g.ex * g.ey * g.ez * g.dof);
    if (g.tile_vol > (unsigned long long)kMaxB) return BIS_OK;
    hipStream_t s = ctx->stream;
    DevBufs B;
    const int grid_n = (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->n_cus * 16);
    // The plan is an optimisation: when the device cannot hold its scratch (two open-addressing tables of 4 E words, the
    // sort buffers) it is "not applicable" -- *out stays null, the error state is cleared and the caller keeps the
    // level-scheduled sweep -- not a failed solve.  (DevBufs and the guard below free what was allocated.)
    auto fail = [&](hipError_t e) {
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); *out = nullptr; return BIS_OK; }
        ctx->err = std::string("tiled sptrsv plan: ") + hipGetErrorString(e);
        return BIS_ERR_HIP;
    };
#define PLAN_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_); } while (0)
#define PLAN_RP(kernel, grid, block, ...) do { \
        if (T->rp64) hipLaunchKernelGGL((kernel<int64_t>), dim3(grid), dim3(block), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((kernel<int32_t>), dim3(grid), dim3(block), 0, s, __VA_ARGS__); \
        PLAN_CHECK(hipGetLastError()); } while (0)
    // the smallest skew under which every operand offset points backwards
    unsigned long long *d_mask = B.get<unsigned long long>(1);
    int *d_bad = B.get<int>(1);
    PLAN_CHECK(B.e);
    PLAN_CHECK(hipMemsetAsync(d_mask, 0xFF, 8, s));
    PLAN_CHECK(hipMemsetAsync(d_bad, 0, 4, s));
    if (T->rp64) hipLaunchKernelGGL(plan_skew_kernel<int64_t>, dim3(grid_n), dim3(256), 0, s, g, n, (const int64_t *)T->row_ptr, T->col, d_mask);
    else hipLaunchKernelGGL(plan_skew_kernel<int32_t>, dim3(grid_n), dim3(256), 0, s, g, n, (const int32_t *)T->row_ptr, T->col, d_mask);
    PLAN_CHECK(hipGetLastError());
    unsigned long long mask = 0;
    PLAN_CHECK(hipMemcpyAsync(&mask, d_mask, 8, hipMemcpyDeviceToHost, s));
    PLAN_CHECK(hipStreamSynchronize(s));
    int sa = -1, sb = -1, sc = -1;
    for (int tot = 0; tot <= 9 && sa < 0; ++tot)
        for (int a2 = 0; a2 <= 3 && sa < 0; ++a2)
            for (int b2 = 0; b2 <= 3 && sa < 0; ++b2) {
                const int c2 = tot - a2 - b2;
                if (c2 < 0 || c2 > 3) continue;
                if (mask >> (a2 + 4 * b2 + 16 * c2) & 1) { sa = a2; sb = b2; sc = c2; }
            }
    if (sa < 0) return BIS_OK;
    g.sa = sa; g.sb = sb; g.sc = sc;
    g.NXs = (g.nx + sa * (g.ny - 1) + sb * (g.nz - 1) + g.ex - 1) / g.ex;
    g.NYs = (g.ny + sc * (g.nz - 1) + g.ey - 1) / g.ey;
    const long long NZs = (g.nz + g.ez - 1) / g.ez;
    g.n_lex = (unsigned long long)(NZs * g.NYs * g.NXs);
    const unsigned long long key_max = (unsigned long long)(NZs + g.NYs + g.NXs) * g.n_lex * g.tile_vol;
    int key_bits = 1;
    while (key_bits < 64 && (key_max >> key_bits) != 0) ++key_bits;
    // order: sort the rows by (hyperplane of tiles, tile, place in the tile)
    unsigned long long *key_in = B.get<unsigned long long>((size_t)n), *key = B.get<unsigned long long>((size_t)n);
    int32_t *row_in = B.get<int32_t>((size_t)n), *ord = B.get<int32_t>((size_t)n), *pos = B.get<int32_t>((size_t)n);
    int32_t *flag = B.get<int32_t>((size_t)n), *tix = B.get<int32_t>((size_t)n);
    int64_t *len = B.get<int64_t>((size_t)n), *entpos = B.get<int64_t>((size_t)n);
    PLAN_CHECK(B.e);
    hipLaunchKernelGGL(plan_key_kernel, dim3(grid_n), dim3(256), 0, s, g, n, key_in, row_in);
    PLAN_CHECK(hipGetLastError());
    {
        size_t b_sort = 0, b_scan32 = 0, b_scan64 = 0;
        PLAN_CHECK(rocprim::radix_sort_pairs(nullptr, b_sort, key_in, key, row_in, ord, (size_t)n, 0, key_bits, s));
        PLAN_CHECK(rocprim::inclusive_scan(nullptr, b_scan32, flag, tix, (size_t)n, rocprim::plus<int32_t>(), s));
        PLAN_CHECK(rocprim::exclusive_scan(nullptr, b_scan64, len, entpos, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), s));
        const size_t b_tmp = std::max(b_sort, std::max(b_scan32, b_scan64));
        char *tmp = B.get<char>(b_tmp);
        PLAN_CHECK(B.e);
        size_t bb = b_tmp;
        PLAN_CHECK(rocprim::radix_sort_pairs(tmp, bb, key_in, key, row_in, ord, (size_t)n, 0, key_bits, s));
        if (T->rp64) hipLaunchKernelGGL(plan_pos_kernel<int64_t>, dim3(grid_n), dim3(256), 0, s, n, g.tile_vol, key, ord, (const int64_t *)T->row_ptr, pos, flag, len);
        else hipLaunchKernelGGL(plan_pos_kernel<int32_t>, dim3(grid_n), dim3(256), 0, s, n, g.tile_vol, key, ord, (const int32_t *)T->row_ptr, pos, flag, len);
        PLAN_CHECK(hipGetLastError());
        bb = b_tmp;
        PLAN_CHECK(rocprim::inclusive_scan(tmp, bb, flag, tix, (size_t)n, rocprim::plus<int32_t>(), s));
        bb = b_tmp;
        PLAN_CHECK(rocprim::exclusive_scan(tmp, bb, len, entpos, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), s));
    }
    if (T->rp64) hipLaunchKernelGGL(plan_verify_kernel<int64_t>, dim3(grid_n), dim3(256), 0, s, n, (const int64_t *)T->row_ptr, T->col, pos, d_bad);
    else hipLaunchKernelGGL(plan_verify_kernel<int32_t>, dim3(grid_n), dim3(256), 0, s, n, (const int32_t *)T->row_ptr, T->col, pos, d_bad);
    PLAN_CHECK(hipGetLastError());
    int32_t n_tiles32 = 0;
    int bad = 0;
    PLAN_CHECK(hipMemcpyAsync(&n_tiles32, tix + (n - 1), 4, hipMemcpyDeviceToHost, s));
    PLAN_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s));
    PLAN_CHECK(hipStreamSynchronize(s));
    if (bad) return BIS_OK; // a pattern the hint does not describe
    const int64_t n_tiles = n_tiles32;
    bis_trsv_tiled *p = new bis_trsv_tiled;
    struct Guard { bis_trsv_tiled *p; ~Guard() { bis_trsv_tiled_destroy(p); } } guard{p};
    p->n = n; p->n_tiles = (int)n_tiles;
    PLAN_CHECK(hipMalloc(&p->tile_slot0, 8 * (size_t)(n_tiles + 1)));
    PLAN_CHECK(hipMalloc(&p->tile_step0, 8 * (size_t)(n_tiles + 1)));
    PLAN_CHECK(hipMalloc(&p->tile_quad0, 8 * (size_t)(n_tiles + 1)));
    PLAN_CHECK(hipMalloc(&p->tile_ext0, 8 * (size_t)(n_tiles + 1)));
    PLAN_CHECK(hipMalloc(&p->slot_row, 4 * (size_t)n));
    PLAN_CHECK(hipMalloc(&p->xs, sizeof(double) * (size_t)(n + 1)));
    PLAN_CHECK(hipMalloc(&p->ticket, sizeof(unsigned) * 4));
    hipLaunchKernelGGL(plan_tiles_kernel, dim3(grid_n), dim3(256), 0, s, n, n_tiles, flag, tix, p->tile_slot0);
    PLAN_CHECK(hipGetLastError());
    // per tile: levels and slots, then count / scan / fill of steps, quads and external ordinals
    a.n = n; a.nnz = T->nnz; a.n_tiles = n_tiles; a.rp = T->row_ptr; a.col = T->col; a.val = T->val;
    a.tile_pos0 = p->tile_slot0; a.ord = ord; a.pos = pos; a.tix = tix; a.entpos = entpos;
    a.lvl = B.get<int32_t>((size_t)n); a.order = B.get<int32_t>((size_t)n); a.lidx = B.get<int32_t>((size_t)n);
    a.cnt = B.get<int32_t>((size_t)(n + 2 * n_tiles)); a.fill = B.get<int32_t>((size_t)(n + 2 * n_tiles));
    a.tile_maxlvl = B.get<int32_t>((size_t)n_tiles);
    a.hash_key = B.get<int32_t>(4 * (size_t)T->nnz + 4); a.hash_ord = B.get<int32_t>(4 * (size_t)T->nnz + 4);
    a.n_demoted = B.get<unsigned long long>(1);
    a.slot_row = p->slot_row;
    a.n_steps = p->tile_step0; a.n_quads = p->tile_quad0; a.n_ext = p->tile_ext0;
    PLAN_CHECK(B.e);
    PLAN_CHECK(hipMemsetAsync(a.n_demoted, 0, 8, s));
    PLAN_CHECK(hipMemsetAsync(a.hash_key, 0xFF, 4 * (4 * (size_t)T->nnz + 4), s));
    PLAN_CHECK(hipMemsetAsync(p->tile_step0, 0, 8 * (size_t)(n_tiles + 1), s));
    PLAN_CHECK(hipMemsetAsync(p->tile_quad0, 0, 8 * (size_t)(n_tiles + 1), s));
    PLAN_CHECK(hipMemsetAsync(p->tile_ext0, 0, 8 * (size_t)(n_tiles + 1), s));
    const int grid_t = (int)((n_tiles + 63) / 64);
    PLAN_RP(plan_levels_kernel, grid_t, 64, a);
    if (T->rp64) hipLaunchKernelGGL((plan_steps_kernel<int64_t, false>), dim3(grid_t), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((plan_steps_kernel<int32_t, false>), dim3(grid_t), dim3(64), 0, s, a);
    PLAN_CHECK(hipGetLastError());
    {
        size_t b_scan = 0;
        PLAN_CHECK(rocprim::exclusive_scan(nullptr, b_scan, p->tile_step0, p->tile_step0, (int64_t)0, (size_t)(n_tiles + 1), rocprim::plus<int64_t>(), s));
        char *tmp = B.get<char>(b_scan);
        PLAN_CHECK(B.e);
        for (int64_t *arr : {p->tile_step0, p->tile_quad0, p->tile_ext0}) {
            size_t bb = b_scan;
            PLAN_CHECK(rocprim::exclusive_scan(tmp, bb, arr, arr, (int64_t)0, (size_t)(n_tiles + 1), rocprim::plus<int64_t>(), s));
        }
    }
    std::vector<int64_t> h_pos0((size_t)n_tiles + 1);
    PLAN_CHECK(hipMemcpyAsync(&p->n_steps, p->tile_step0 + n_tiles, 8, hipMemcpyDeviceToHost, s));
    PLAN_CHECK(hipMemcpyAsync(&p->n_quads, p->tile_quad0 + n_tiles, 8, hipMemcpyDeviceToHost, s));
    PLAN_CHECK(hipMemcpyAsync(&p->n_ext, p->tile_ext0 + n_tiles, 8, hipMemcpyDeviceToHost, s));
    PLAN_CHECK(hipMemcpyAsync(h_pos0.data(), p->tile_slot0, 8 * h_pos0.size(), hipMemcpyDeviceToHost, s));
    PLAN_CHECK(hipStreamSynchronize(s));
    for (int64_t t = 0; t < n_tiles; ++t) p->max_rows = std::max(p->max_rows, (int)(h_pos0[(size_t)t + 1] - h_pos0[(size_t)t]));
    PLAN_CHECK(hipMalloc(&p->step_desc, sizeof(int4) * (size_t)std::max<int64_t>(p->n_steps, 1)));
    PLAN_CHECK(hipMalloc(&p->quad_code, sizeof(int4) * (size_t)std::max<int64_t>(p->n_quads, 1)));
    PLAN_CHECK(hipMalloc(&p->quad_val, sizeof(double) * 4 * (size_t)std::max<int64_t>(p->n_quads, 1)));
    PLAN_CHECK(hipMalloc(&p->ext_src, 4 * (size_t)std::max<int64_t>(p->n_ext, 4)));
    a.step_desc = p->step_desc; a.quad_code = p->quad_code; a.quad_val = (double *)p->quad_val; a.ext_src = p->ext_src;
    PLAN_CHECK(hipMemsetAsync(a.hash_key, 0xFF, 4 * (4 * (size_t)T->nnz + 4), s));
    if (T->rp64) hipLaunchKernelGGL((plan_steps_kernel<int64_t, true>), dim3(grid_t), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((plan_steps_kernel<int32_t, true>), dim3(grid_t), dim3(64), 0, s, a);
    PLAN_CHECK(hipGetLastError());
    unsigned long long n_demoted = 0;
    PLAN_CHECK(hipMemcpyAsync(&n_demoted, a.n_demoted, 8, hipMemcpyDeviceToHost, s));
    PLAN_CHECK(hipStreamSynchronize(s));
#undef PLAN_RP
#undef PLAN_CHECK
    if (getenv("BIS_TRSV_TILE_STATS"))
        fprintf(stderr, "tiled sptrsv plan (%s): %lld rows, %d tiles (largest %d rows), %lld steps (%.1f rows per step), %lld quads (%.2f x the entries), "
                        "%lld external ordinals (%.2f per row), %lld in-tile operands beyond the ring; grid tiles, skew %d %d %d; device plan\n",
                backward ? "backward" : "forward", (long long)n, p->n_tiles, p->max_rows, (long long)p->n_steps,
                (double)n / (double)std::max<int64_t>(p->n_steps, 1), (long long)p->n_quads, 4.0 * (double)p->n_quads / (double)T->nnz,
                (long long)p->n_ext, (double)p->n_ext / (double)n, (long long)n_demoted, sa, sb, sc);
    guard.p = nullptr;
    *out = p;
    return BIS_OK;
}

// trsv_tiled: -1 (default) the tiled sweep where the device plan applies (a valid grid hint); 1 also elsewhere (host
// plan: interval tiles, seconds at 10^7 rows); 2 host plan always (the device plan's oracle in the tests); 0 never
bis_status bis_trsv_tiled_build(bis_ctx *ctx, const bis_mat *T, bool backward, bis_trsv_tiled **out) {
    *out = nullptr;
    const int mode = bis_opts().trsv_tiled;
    if (mode == 0) return BIS_OK;
    if (mode != 2) {
        const bis_status st = trsv_tiled_build_device(ctx, T, backward, out);
        if (st != BIS_OK || *out) return st;
    }
    return mode > 0 ? trsv_tiled_build_host(ctx, T, backward, out) : BIS_OK;
}

bis_status bis_trsv_tiled_solve(bis_ctx *ctx, bis_trsv_tiled *p, double *x, const double *D, const double *b) {
    const int fill_grid = (int)std::min<int64_t>((p->n + 1 + 255) / 256, 2048);
    hipLaunchKernelGGL(fill_sentinel_kernel, dim3(fill_grid), dim3(256), 0, ctx->stream, p->xs, p->n + 1);
    BIS_HIP_CHECK(ctx, hipMemsetAsync(p->ticket, 0, sizeof(unsigned) * 4, ctx->stream));
    TiledArgs a{p->slot_row, p->step_desc, p->tile_slot0, p->tile_step0, p->tile_quad0, p->tile_ext0, p->quad_code, p->quad_val,
                p->ext_src, p->xs, p->ticket, D, b, x, ctx->fault_dev, ctx->spmv_stop, p->n_tiles, bis_opts().trsv_tile_backoff >= 0 ? std::min(bis_opts().trsv_tile_backoff, 64) : 16, std::max(bis_opts().trsv_tile_exp, 0), nullptr, nullptr, nullptr, nullptr};
    static long long *dbg_buf = nullptr; // diagnostic (BIS_TRSV_TILE_DEBUG=file): per-tile stamps of the last sweep
    static int64_t dbg_cap = 0;
    const char *dbg_file = getenv("BIS_TRSV_TILE_DEBUG");
    if (dbg_file) {
        // layout: 16 words per tile, then the publish stamp of every slot, then the delivery stamp of every external ordinal
        const int64_t words = 16 * (int64_t)p->n_tiles + p->n + 1 + p->n_ext;
        if (dbg_cap < words) { hipFree(dbg_buf); hipMalloc(&dbg_buf, sizeof(long long) * (size_t)words); dbg_cap = words; }
        hipMemsetAsync(dbg_buf, 0, sizeof(long long) * (size_t)words, ctx->stream);
        a.dbg = dbg_buf;
        static long long *dbg_step = nullptr;
        if (!dbg_step) hipMalloc(&dbg_step, sizeof(long long) * 512);
        hipMemsetAsync(dbg_step, 0, sizeof(long long) * 512, ctx->stream);
        a.dbg_step = dbg_step;
        a.dbg_pub = dbg_buf + 16 * (int64_t)p->n_tiles;
        a.dbg_del = a.dbg_pub + p->n + 1;
    }
    // resident workgroups per CU: what the occupancy query says (3: the 51 KiB of LDS and the ~135 VGPRs both allow exactly that)
    static int res = 0;
    const bool exp = a.exp_flags != 0;
    if (res == 0) {
        int nb = 0;
        const hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trsv_tiled_kernel<Cfg, false>, 256, 0);
        res = (oe == hipSuccess && nb > 0) ? std::min(nb, 8) : 1;
        (void)hipGetLastError();
        if (getenv("BIS_TRSV_TILE_STATS")) fprintf(stderr, "tiled sptrsv: %d workgroups per CU resident\n", res);
    }
    const int per_cu = bis_opts().trsv_tile_wgs > 0 ? std::min(bis_opts().trsv_tile_wgs, res) : res;
    const int grid = (int)std::min<int64_t>(p->n_tiles, (int64_t)ctx->n_cus * per_cu);
    // The timing-experiment build of the kernel (exp_flags: results are WRONG with any bit but 64 set) and the stamped debug build
    // exist only in a library compiled with -DBIS_TILED_EXP (tools/trsv_ab.py, tools/trsv_tile_debug.py: BIS_EXTRA_HIPCC_FLAGS);
    // the product refuses the option instead of returning wrong numbers silently.
#ifdef BIS_TILED_EXP
    if (dbg_file) hipLaunchKernelGGL((trsv_tiled_kernel<Cfg, true, true>), dim3(grid), dim3(256), 0, ctx->stream, a);
    else if (exp) hipLaunchKernelGGL((trsv_tiled_kernel<Cfg, false, true>), dim3(grid), dim3(256), 0, ctx->stream, a);
    else
#else
    if (dbg_file || exp) {
        ctx->err = "tiled sptrsv: trsv_tile_exp / BIS_TRSV_TILE_DEBUG need a library built with -DBIS_TILED_EXP (timing experiments: wrong results by design)";
        return BIS_ERR_INVALID;
    }
#endif
    hipLaunchKernelGGL((trsv_tiled_kernel<Cfg, false>), dim3(grid), dim3(256), 0, ctx->stream, a);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    if (dbg_file) {
        // file: {tiles, slots, external ordinals}, the stamps, then the producing slot of every ordinal and the tiles' first ordinals
        const int64_t words = 16 * (int64_t)p->n_tiles + p->n + 1 + p->n_ext;
        std::vector<long long> h((size_t)words);
        std::vector<int32_t> src((size_t)p->n_ext);
        std::vector<int64_t> e0((size_t)p->n_tiles + 1), s0((size_t)p->n_tiles + 1);
        hipStreamSynchronize(ctx->stream);
        hipMemcpy(h.data(), dbg_buf, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        hipMemcpy(src.data(), p->ext_src, 4 * src.size(), hipMemcpyDeviceToHost);
        hipMemcpy(e0.data(), p->tile_ext0, 8 * e0.size(), hipMemcpyDeviceToHost);
        hipMemcpy(s0.data(), p->tile_slot0, 8 * s0.size(), hipMemcpyDeviceToHost);
        if (FILE *f = fopen(dbg_file, "wb")) {
            const long long head[3] = {p->n_tiles, p->n + 1, p->n_ext};
            fwrite(head, 8, 3, f);
            fwrite(h.data(), sizeof(long long), h.size(), f);
            fwrite(src.data(), 4, src.size(), f);
            fwrite(e0.data(), 8, e0.size(), f);
            fwrite(s0.data(), 8, s0.size(), f);
            long long hs[512];
            hipMemcpy(hs, a.dbg_step, sizeof(hs), hipMemcpyDeviceToHost);
            fwrite(hs, 8, 512, f);
            fclose(f);
        }
    }
    return BIS_OK;
}
