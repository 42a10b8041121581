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

struct bis_trsv_tiled {
    int64_t n = 0;
    int B = 0;                 // rows per tile
    int n_tiles = 0;
    int64_t n_steps = 0, n_ent = 0, n_ext = 0;
    // device arrays
    int2 *slot_rowlen = nullptr;    // [n]      slot -> {row, number of entries}; slots = rows in (tile, local level, order) order
    int2 *step_desc = nullptr;      // [n_steps + n_tiles] per tile: {first slot | longest row << 16, first entry} of each step, relative to the tile, + end marker
    int64_t *tile_step0 = nullptr;  // [n_tiles + 1] index into step_desc (tile t owns [tile_step0[t], tile_step0[t+1]) incl. its end marker)
    int64_t *tile_ent0 = nullptr;   // [n_tiles + 1]
    int64_t *tile_ext0 = nullptr;   // [n_tiles + 1]
    double *ent_val = nullptr;      // [n_ent]  entries per step, k-major / lane-minor (padded to the step's longest row)
    int32_t *ent_code = nullptr;    // [n_ent]  operand index: < B the tile's own slot, else B + external ordinal
    int32_t *ext_src = nullptr;     // [n_ext]  slot (global) whose published result the ordinal stands for
    unsigned long long *xs = nullptr; // [n + 1] published results, by slot
    unsigned *ticket = nullptr;
};

void bis_trsv_tiled_destroy(bis_trsv_tiled *p) {
    if (!p) return;
    hipFree(p->slot_rowlen); hipFree(p->step_desc); hipFree(p->tile_step0); hipFree(p->tile_ent0);
    hipFree(p->tile_ext0); hipFree(p->ent_val); hipFree(p->ent_code); hipFree(p->ext_src);
    hipFree(p->xs); hipFree(p->ticket);
    delete p;
}

namespace {

constexpr unsigned long long kSentinel = 0x7FF85EA71E55C0DEull; // quiet NaN + payload (same as bis_sptrsv.hip)
constexpr unsigned long long kCanonNaN = 0x7FF8000000000000ull;
constexpr int kMaxB = 2048;       // rows per tile
constexpr int kOpnd = 4096;       // operands of a tile in LDS (32 KiB): its own B results + its distinct external operands
constexpr int kRingEnt = 2048;    // entry ring (LDS: 16 + 8 KiB)
constexpr int kRingSlot = 256;    // per-row operand ring (LDS: 2 + 4 KiB)
constexpr int kEntChunk = 1024;   // entries per loader round (16 per lane)
constexpr int kSlotChunk = 128;   // rows per loader round (2 per lane)
constexpr unsigned kSpinLds = 1u << 24;  // polls of an LDS word before a wave gives up (seconds)
constexpr unsigned kSpinMem = 1u << 22;  // polls of a memory word

enum { C_TICKET = 0, C_ENT_LOADED, C_ENT_DONE, C_SLOT_LOADED, C_SLOT_DONE, C_N = 8 };

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

__global__ __launch_bounds__(256) void fill_sentinel_kernel(unsigned long long *xs, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) xs[i] = kSentinel;
}

// ent_val[j] = val[src[j]] (src < 0: padding)
template <typename IX>
__global__ __launch_bounds__(256) void gather_entries_kernel(const double *__restrict__ val, const IX *__restrict__ src,
                                                             int64_t n, double *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) {
        const IX k = src[j];
        out[j] = k >= 0 ? val[k] : 0.0;
    }
}

struct TiledArgs {
    const int2 *slot_rowlen;
    const int2 *step_desc;
    const int64_t *tile_step0, *tile_ent0, *tile_ext0;
    const double *ent_val;
    const int32_t *ent_code;
    const int32_t *ext_src;
    unsigned long long *xs;
    unsigned *ticket;
    const double *D, *b;
    double *x;
    unsigned *fault;
    int64_t n;
    int B, n_tiles;
    long long *dbg; // optional, 8 words per tile: start, end (s_memtime), cycles the compute wave waited for the loaders /
                    // for external operands, end of the entry loader / slot loader / poller, steps
};

__global__ __launch_bounds__(256) void trsv_tiled_kernel(const TiledArgs a) {
    // operands of the tile's rows: [0, B) results of the tile itself, by slot; [B, kOpnd) the
    // external operands, by ordinal (sentinel until the poller delivers them).  The entry codes index this
    // array directly: one LDS read per operand, no branch on where it comes from.
    __shared__ unsigned long long opnd[kOpnd];
    __shared__ double ring_val[kRingEnt];
    __shared__ int ring_code[kRingEnt];
    __shared__ int2 ring_rowlen[kRingSlot];
    __shared__ double2 ring_bD[kRingSlot];
    __shared__ unsigned ctl[C_N];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (;;) {
        if (threadIdx.x == 0) ctl[C_TICKET] = atomicAdd(a.ticket, 1u);
        if (threadIdx.x >= 1 && threadIdx.x < C_N) ctl[threadIdx.x] = 0u;
        __syncthreads();
        const int t = (int)ctl[C_TICKET];
        if (t >= a.n_tiles) return; // every wave reaches this once the tickets run out
        const int64_t slot0 = (int64_t)t * a.B;
        const int n_slots = (int)min((int64_t)a.B, a.n - slot0);
        const int64_t sd0 = a.tile_step0[t];
        const int n_steps = (int)(a.tile_step0[t + 1] - sd0) - 1;
        const int64_t ent0 = a.tile_ent0[t];
        const int n_ent = (int)(a.tile_ent0[t + 1] - ent0);
        const int64_t ext0 = a.tile_ext0[t];
        const int n_ext = (int)(a.tile_ext0[t + 1] - ext0);
        for (int e = threadIdx.x; e < n_ext; e += 256) opnd[a.B + e] = kSentinel;
        __syncthreads();

        const long long t_start = a.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
        long long w_load = 0, w_ext = 0;
        if (wave == 0) {
            // ---- compute wave: one lane per row of the step, CRS-order fma chain ----
            // Per step the dependent chain is: operand read (LDS) -> fma chain -> division -> result write
            // (LDS).  Everything else is taken off it: the step's codes / values / b / D are read from the
            // rings while the PREVIOUS step divides (they do not depend on its result), the loaders'
            // watermarks are re-read only when the cached copy does not cover the step, and the wave's own
            // progress is published every 8 steps.
            int2 d_cur = a.step_desc[sd0 + min(lane, n_steps)];
            int2 d_nxt = a.step_desc[sd0 + min(lane + 1, n_steps)];
            int2 p_cur = d_cur, p_nxt = d_nxt; // the following batch of 64 step descriptors, fetched a batch ahead
            int ent_loaded = 0, slot_loaded = 0; // cached watermarks of the loaders
            bool have = false;                   // the ring reads of the coming step are already in flight
            int2 rl = make_int2(0, 0);
            double2 bd = make_double2(0.0, 1.0);
            int code[4] = {0, 0, 0, 0};
            double v[4] = {0.0, 0.0, 0.0, 0.0};
            for (int s = 0; s < n_steps; ++s) {
                const int j = s & 63;
                if (j == 0) {
                    if (s > 0) { d_cur = p_cur; d_nxt = p_nxt; }
                    p_cur = a.step_desc[sd0 + min(s + 64 + lane, n_steps)];
                    p_nxt = a.step_desc[sd0 + min(s + 65 + lane, n_steps)];
                }
                const int dx = __builtin_amdgcn_readlane(d_cur.x, j), ent_b = __builtin_amdgcn_readlane(d_cur.y, j);
                const int dxe = __builtin_amdgcn_readlane(d_nxt.x, j), ent_e = __builtin_amdgcn_readlane(d_nxt.y, j);
                const int slot_b = dx & 0xffff, L = dx >> 16, slot_e = dxe & 0xffff;
                const int w = slot_e - slot_b;
                const bool active = lane < w;
                if (!have) {
                    // the loaders have to be past this step
                    unsigned spins = 0;
                    const long long t0 = a.dbg ? (long long)__builtin_readcyclecounter() : 0;
                    if (ent_loaded < ent_e || slot_loaded < slot_e) { // about to wait: the loaders must know how far the rings are free
                        if (lane == 0) { lds_release(&ctl[C_ENT_DONE], (unsigned)ent_b); lds_release(&ctl[C_SLOT_DONE], (unsigned)slot_b); }
                    }
                    while (ent_loaded < ent_e || slot_loaded < slot_e) {
                        ent_loaded = (int)lds_acquire(&ctl[C_ENT_LOADED]);
                        slot_loaded = (int)lds_acquire(&ctl[C_SLOT_LOADED]);
                        if (ent_loaded >= ent_e && slot_loaded >= slot_e) break;
                        if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (a.dbg) w_load += (long long)__builtin_readcyclecounter() - t0;
                    const int sl = (slot_b + lane) & (kRingSlot - 1);
                    rl = active ? ring_rowlen[sl] : make_int2(0, 0);
                    bd = active ? ring_bD[sl] : make_double2(0.0, 1.0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int idx = (ent_b + q * w + lane) & (kRingEnt - 1);
                        code[q] = active && q < L ? ring_code[idx] : 0;
                        v[q] = active && q < L ? ring_val[idx] : 0.0;
                    }
                }
                have = false;
                double acc = 0.0;
                bool lost = false;
                // rounds of 4 entries: their operands are read together; the fma chain then runs in CRS order
                for (int k0 = 0; k0 < L; k0 += 4) {
                    unsigned long long bits[4];
                    if (k0 > 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int idx = (ent_b + (k0 + q) * w + lane) & (kRingEnt - 1);
                            code[q] = active && k0 + q < L ? ring_code[idx] : 0;
                            v[q] = active && k0 + q < L ? ring_val[idx] : 0.0;
                        }
                    }
                    bool on[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        on[q] = active && k0 + q < rl.y;
                        bits[q] = __hip_atomic_load(&opnd[code[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    // wait (whole wave, LDS only) until the poller has delivered the external operands of this round
                    unsigned sp2 = 0;
                    const long long t1 = a.dbg ? (long long)__builtin_readcyclecounter() : 0;
                    for (;;) {
                        bool pend = false;
#pragma unroll
                        for (int q = 0; q < 4; ++q) pend |= on[q] && bits[q] == kSentinel; // results of the tile are never the sentinel
                        if (!__ballot(pend)) break;
                        if (++sp2 > kSpinLds) { lost = true; break; }
                        __builtin_amdgcn_s_sleep(1);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (on[q] && bits[q] == kSentinel)
                                bits[q] = __hip_atomic_load(&opnd[code[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    if (a.dbg) w_ext += (long long)__builtin_readcyclecounter() - t1;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (on[q]) acc = fma(v[q], __longlong_as_double((long long)bits[q]), acc);
                }
                const double num = bd.x - acc, den = bd.y;
                const int row = rl.x;
                // ring reads of the next step, issued before the division of this one (same descriptor batch, and
                // the cached watermarks already cover it; otherwise the next trip does it the slow way)
                if (j < 63 && s + 1 < n_steps) {
                    const int n_dx = __builtin_amdgcn_readlane(d_cur.x, j + 1), n_ent_b = ent_e;
                    const int n_dxe = __builtin_amdgcn_readlane(d_nxt.x, j + 1), n_ent_e = __builtin_amdgcn_readlane(d_nxt.y, j + 1);
                    const int n_slot_b = n_dx & 0xffff, n_L = n_dx >> 16, n_slot_e = n_dxe & 0xffff;
                    if (ent_loaded >= n_ent_e && slot_loaded >= n_slot_e) {
                        const int n_w = n_slot_e - n_slot_b;
                        const bool n_active = lane < n_w;
                        const int sl = (n_slot_b + lane) & (kRingSlot - 1);
                        rl = n_active ? ring_rowlen[sl] : make_int2(0, 0);
                        bd = n_active ? ring_bD[sl] : make_double2(0.0, 1.0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int idx = (n_ent_b + q * n_w + lane) & (kRingEnt - 1);
                            code[q] = n_active && q < n_L ? ring_code[idx] : 0;
                            v[q] = n_active && q < n_L ? ring_val[idx] : 0.0;
                        }
                        have = true;
                    }
                }
                const double res = num / den;
                unsigned long long out = (unsigned long long)__double_as_longlong(res);
                if (res != res || lost) out = kCanonNaN; // never publish the sentinel pattern
                if (lost && lane == 0) __hip_atomic_fetch_or(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (active) {
                    __hip_atomic_store(&opnd[slot_b + lane], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    a.x[row] = __longlong_as_double((long long)out);
                    __hip_atomic_store(&a.xs[slot0 + slot_b + lane], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (lane == 0 && ((s & 7) == 7 || s + 1 == n_steps)) {
                    // what the rings may overwrite: everything before this step's successor (its reads may be in flight)
                    lds_release(&ctl[C_ENT_DONE], (unsigned)ent_b);
                    lds_release(&ctl[C_SLOT_DONE], (unsigned)slot_b);
                }
            }
            if (a.dbg && lane == 0) {
                long long *d = a.dbg + (int64_t)t * 8;
                d[0] = t_start; d[1] = (long long)__builtin_amdgcn_s_memrealtime(); d[2] = w_load; d[3] = w_ext; d[7] = n_steps;
            }
        } else if (wave == 1) {
            // ---- entry loader: the tile's entry stream (step order, k-major / lane-minor) into the ring ----
            for (int done = 0; done < n_ent;) {
                const int chunk = min(kEntChunk, n_ent - done);
                unsigned spins = 0;
                while (done + chunk - (int)lds_acquire(&ctl[C_ENT_DONE]) > kRingEnt) {
                    if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                double v[kEntChunk / 64];
                int c[kEntChunk / 64];
#pragma unroll
                for (int u = 0; u < kEntChunk / 64; ++u) {
                    const int i = u * 64 + lane;
                    const int64_t g = ent0 + done + min(i, chunk - 1);
                    v[u] = a.ent_val[g];
                    c[u] = a.ent_code[g];
                }
#pragma unroll
                for (int u = 0; u < kEntChunk / 64; ++u) {
                    const int i = u * 64 + lane;
                    if (i < chunk) {
                        ring_val[(done + i) & (kRingEnt - 1)] = v[u];
                        ring_code[(done + i) & (kRingEnt - 1)] = c[u];
                    }
                }
                done += chunk;
                lds_release(&ctl[C_ENT_LOADED], (unsigned)done);
            }
            if (a.dbg && lane == 0) a.dbg[(int64_t)t * 8 + 4] = (long long)__builtin_amdgcn_s_memrealtime();
        } else if (wave == 2) {
            // ---- per-row operand loader: row index, row length, b[row], D[row] in slot order ----
            constexpr int U = kSlotChunk / 64;
            int2 rl_next[U];
#pragma unroll
            for (int u = 0; u < U; ++u) rl_next[u] = a.slot_rowlen[slot0 + min(u * 64 + lane, n_slots - 1)];
            for (int done = 0; done < n_slots;) {
                const int chunk = min(kSlotChunk, n_slots - done);
                int2 rl[U];
                double bv[U], dv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    rl[u] = rl_next[u];
                    bv[u] = a.b[rl[u].x];
                    dv[u] = a.D[rl[u].x];
                }
                if (done + chunk < n_slots) {
#pragma unroll
                    for (int u = 0; u < U; ++u) rl_next[u] = a.slot_rowlen[slot0 + min(done + chunk + u * 64 + lane, n_slots - 1)];
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
                        ring_rowlen[(done + i) & (kRingSlot - 1)] = rl[u];
                        ring_bD[(done + i) & (kRingSlot - 1)] = make_double2(bv[u], dv[u]);
                    }
                }
                done += chunk;
                lds_release(&ctl[C_SLOT_LOADED], (unsigned)done);
            }
            if (a.dbg && lane == 0) a.dbg[(int64_t)t * 8 + 5] = (long long)__builtin_amdgcn_s_memrealtime();
        } else {
            // ---- poller: external operands in first-need order; every lane advances on its own ----
            int e = lane;
            int src = e < n_ext ? a.ext_src[ext0 + e] : 0;
            int src_next = e + 64 < n_ext ? a.ext_src[ext0 + e + 64] : 0;
            unsigned spins = 0;
            while (__ballot(e < n_ext)) {
                if (e < n_ext) {
                    const unsigned long long vbits = __hip_atomic_load(&a.xs[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool give_up = ++spins > kSpinMem;
                    if (vbits != kSentinel || give_up) {
                        if (give_up) __hip_atomic_fetch_or(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(&opnd[a.B + e], give_up && vbits == kSentinel ? kCanonNaN : vbits, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                        e += 64;
                        src = src_next;
                        src_next = e + 64 < n_ext ? a.ext_src[ext0 + e + 64] : 0;
                        spins = 0;
                    }
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (a.dbg && lane == 0) a.dbg[(int64_t)t * 8 + 6] = (long long)__builtin_amdgcn_s_memrealtime();
        }
        __syncthreads();
    }
}

} // namespace

// ---- plan (host analysis, version 1) --------------------------------------------------------------
// Input: the strictly triangular pattern on the host.  Processing order: ascending rows (forward) or
// descending rows (backward) -- the reference's substitution order, always a linear extension.
bis_status bis_trsv_tiled_build(bis_ctx *ctx, const bis_mat *T, bool backward, bis_trsv_tiled **out) {
    *out = nullptr;
    const int64_t n = T->n_rows;
    if (n == 0 || T->nnz == 0 || T->nnz > (int64_t)600000000 || T->view) return BIS_OK; // not applicable: caller keeps the level-scheduled sweep
    std::vector<int64_t> rp(n + 1);
    std::vector<int32_t> col((size_t)T->nnz);
    bis_status st = bis_mat_download(ctx, T, rp.data(), col.data(), nullptr);
    if (st != BIS_OK) return st;
    auto row_at = [&](int64_t pos) { return backward ? n - 1 - pos : pos; };
    auto pos_of = [&](int64_t row) { return backward ? n - 1 - row : row; };
    int max_len = 0;
    for (int64_t r = 0; r < n; ++r) max_len = std::max<int>(max_len, (int)(rp[r + 1] - rp[r]));
    if (max_len > kRingEnt / 2) return BIS_OK; // a single row must fit half the entry ring
    int B = bis_opts().trsv_tile_rows > 0 ? std::min(bis_opts().trsv_tile_rows, kMaxB) : kMaxB;
    for (;; B /= 2) {
        if (B < 64) return BIS_OK; // tiles would be too small to pay: keep the level-scheduled sweep
        const int64_t n_tiles = (n + B - 1) / B;
        if (n_tiles > INT32_MAX) return BIS_OK;
        std::vector<int2> slot_rowlen((size_t)n);
        std::vector<int2> step_desc;
        std::vector<int64_t> tile_step0(n_tiles + 1, 0), tile_ent0(n_tiles + 1, 0), tile_ext0(n_tiles + 1, 0);
        std::vector<int32_t> ent_code, ext_src;
        std::vector<int64_t> ent_src; // CRS index of each entry, -1 = padding
        std::vector<int32_t> lidx((size_t)n); // row -> slot within its tile
        step_desc.reserve((size_t)(n / 8));
        ent_code.reserve((size_t)(T->nnz + T->nnz / 8));
        ent_src.reserve((size_t)(T->nnz + T->nnz / 8));
        std::vector<int> lvl(B), order(B), cnt;
        std::vector<int32_t> ext_stamp((size_t)n, -1), ext_ord((size_t)n, 0); // per global slot: tile that listed it last, its ordinal there
        bool too_many_ext = false;
        for (int64_t t = 0; t < n_tiles && !too_many_ext; ++t) {
            const int64_t p0 = t * B;
            const int m = (int)std::min<int64_t>(B, n - p0);
            // local levels: in-tile dependencies only (they sit at earlier positions of the tile)
            int max_lvl = 0;
            for (int i = 0; i < m; ++i) {
                const int64_t r = row_at(p0 + i);
                int l = 0;
                for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
                    const int64_t q = pos_of(col[k]);
                    if (q >= p0) l = std::max(l, lvl[(int)(q - p0)] + 1);
                }
                lvl[i] = l;
                max_lvl = std::max(max_lvl, l);
            }
            // stable counting sort by level
            cnt.assign(max_lvl + 2, 0);
            for (int i = 0; i < m; ++i) cnt[lvl[i] + 1]++;
            for (int l = 0; l <= max_lvl; ++l) cnt[l + 1] += cnt[l];
            {
                std::vector<int> fill(cnt.begin(), cnt.end() - 1);
                for (int i = 0; i < m; ++i) order[fill[lvl[i]]++] = i;
            }
            for (int s = 0; s < m; ++s) {
                const int64_t r = row_at(p0 + order[s]);
                slot_rowlen[(size_t)(p0 + s)] = make_int2((int)r, (int)(rp[r + 1] - rp[r]));
                lidx[(size_t)r] = s;
            }
            // steps: runs of one level, at most 64 rows, at most kRingEnt/2 padded entries
            tile_step0[t] = (int64_t)step_desc.size();
            tile_ent0[t] = (int64_t)ent_code.size();
            tile_ext0[t] = (int64_t)ext_src.size();
            int n_ext_tile = 0;
            for (int l = 0; l <= max_lvl; ++l) {
                int s = cnt[l];
                const int s_end = cnt[l + 1];
                while (s < s_end) {
                    int w = 0, L = 0;
                    while (s + w < s_end && w < 64) {
                        const int len = slot_rowlen[(size_t)(p0 + s + w)].y;
                        const int L2 = std::max(L, len);
                        if (w > 0 && (int64_t)(w + 1) * L2 > kRingEnt / 2) break;
                        L = L2;
                        ++w;
                    }
                    step_desc.push_back(make_int2(s | (L << 16), (int)((int64_t)ent_code.size() - tile_ent0[t])));
                    for (int k = 0; k < L; ++k)
                        for (int i = 0; i < w; ++i) {
                            const int2 rl = slot_rowlen[(size_t)(p0 + s + i)];
                            if (k >= rl.y) { ent_code.push_back(0); ent_src.push_back(-1); continue; }
                            const int64_t kk = rp[rl.x] + k;
                            const int64_t q = pos_of(col[kk]);
                            ent_src.push_back(kk);
                            if (q >= p0) {
                                ent_code.push_back(lidx[(size_t)col[kk]]);
                            } else {
                                const int32_t gs = (int32_t)((q / B) * B + lidx[(size_t)col[kk]]);
                                if (ext_stamp[(size_t)gs] != (int32_t)t) { // first need in this tile: next ordinal
                                    ext_stamp[(size_t)gs] = (int32_t)t;
                                    ext_ord[(size_t)gs] = n_ext_tile++;
                                    ext_src.push_back(gs);
                                }
                                ent_code.push_back(B + ext_ord[(size_t)gs]);
                            }
                        }
                    s += w;
                }
            }
            step_desc.push_back(make_int2(m, (int)((int64_t)ent_code.size() - tile_ent0[t]))); // end marker
            if (n_ext_tile > kOpnd - B) too_many_ext = true;
        }
        if (too_many_ext) continue; // halve the tile
        tile_step0[n_tiles] = (int64_t)step_desc.size();
        tile_ent0[n_tiles] = (int64_t)ent_code.size();
        tile_ext0[n_tiles] = (int64_t)ext_src.size();
        // upload
        bis_trsv_tiled *p = new bis_trsv_tiled;
        p->n = n; p->B = B; p->n_tiles = (int)n_tiles;
        p->n_steps = (int64_t)step_desc.size() - n_tiles;
        p->n_ent = (int64_t)ent_code.size();
        p->n_ext = (int64_t)ext_src.size();
        int64_t *d_src = nullptr;
        hipError_t e = hipMalloc(&p->slot_rowlen, sizeof(int2) * (size_t)n);
        auto up = [&](void **dst, const void *src, size_t bytes) {
            if (e != hipSuccess) return;
            e = hipMalloc(dst, std::max<size_t>(bytes, 16));
            if (e == hipSuccess && bytes) e = hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
        };
        if (e == hipSuccess) e = hipMemcpyAsync(p->slot_rowlen, slot_rowlen.data(), sizeof(int2) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
        up((void **)&p->step_desc, step_desc.data(), sizeof(int2) * step_desc.size());
        up((void **)&p->tile_step0, tile_step0.data(), 8 * tile_step0.size());
        up((void **)&p->tile_ent0, tile_ent0.data(), 8 * tile_ent0.size());
        up((void **)&p->tile_ext0, tile_ext0.data(), 8 * tile_ext0.size());
        up((void **)&p->ent_code, ent_code.data(), 4 * ent_code.size());
        up((void **)&p->ext_src, ext_src.data(), 4 * ext_src.size());
        up((void **)&d_src, ent_src.data(), 8 * ent_src.size());
        if (e == hipSuccess) e = hipMalloc(&p->ent_val, sizeof(double) * std::max<size_t>(ent_code.size(), 2));
        if (e == hipSuccess) e = hipMalloc(&p->xs, sizeof(double) * (size_t)(n + 1));
        if (e == hipSuccess) e = hipMalloc(&p->ticket, sizeof(unsigned) * 4);
        if (e == hipSuccess && p->n_ent > 0) {
            hipLaunchKernelGGL(gather_entries_kernel<int64_t>, dim3((unsigned)std::min<int64_t>((p->n_ent + 255) / 256, 8192)), dim3(256), 0,
                               ctx->stream, T->val, d_src, p->n_ent, p->ent_val);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        hipFree(d_src);
        if (e != hipSuccess) {
            ctx->err = std::string("tiled sptrsv plan: ") + hipGetErrorString(e);
            bis_trsv_tiled_destroy(p);
            return BIS_ERR_HIP;
        }
        *out = p;
        return BIS_OK;
    }
}

// the matrix' values changed in place (bis_mat_scale_sym on a triangle is not a thing today, but ILU
// refactorisation could): re-gather is the caller's business -- plans are built per matrix and dropped with it.

bis_status bis_trsv_tiled_solve(bis_ctx *ctx, bis_trsv_tiled *p, double *x, const double *D, const double *b) {
    const int fill_grid = (int)std::min<int64_t>((p->n + 1 + 255) / 256, 2048);
    hipLaunchKernelGGL(fill_sentinel_kernel, dim3(fill_grid), dim3(256), 0, ctx->stream, p->xs, p->n + 1);
    BIS_HIP_CHECK(ctx, hipMemsetAsync(p->ticket, 0, sizeof(unsigned) * 4, ctx->stream));
    TiledArgs a{p->slot_rowlen, p->step_desc, p->tile_step0, p->tile_ent0, p->tile_ext0, p->ent_val, p->ent_code,
                p->ext_src, p->xs, p->ticket, D, b, x, ctx->fault_dev, p->n, p->B, p->n_tiles, nullptr};
    static long long *dbg_buf = nullptr; // diagnostic (BIS_TRSV_TILE_DEBUG=file): per-tile cycle stamps of the last sweep
    static int64_t dbg_cap = 0;
    const char *dbg_file = getenv("BIS_TRSV_TILE_DEBUG");
    if (dbg_file) {
        if (dbg_cap < p->n_tiles) { hipFree(dbg_buf); hipMalloc(&dbg_buf, sizeof(long long) * 8 * (size_t)p->n_tiles); dbg_cap = p->n_tiles; }
        hipMemsetAsync(dbg_buf, 0, sizeof(long long) * 8 * (size_t)p->n_tiles, ctx->stream);
        a.dbg = dbg_buf;
    }
    int per_cu = bis_opts().trsv_tile_wgs > 0 ? bis_opts().trsv_tile_wgs : 2;
    const int grid = (int)std::min<int64_t>(p->n_tiles, (int64_t)ctx->n_cus * per_cu);
    hipLaunchKernelGGL(trsv_tiled_kernel, dim3(grid), dim3(256), 0, ctx->stream, a);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    if (dbg_file) {
        std::vector<long long> h((size_t)p->n_tiles * 8);
        hipStreamSynchronize(ctx->stream);
        hipMemcpy(h.data(), dbg_buf, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        if (FILE *f = fopen(dbg_file, "wb")) { fwrite(h.data(), sizeof(long long), h.size(), f); fclose(f); }
    }
    return BIS_OK;
}
