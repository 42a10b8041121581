// bis_trsv_chain.hip -- natural-order sparse triangular sweeps (reference kernels.hpp:54-117, serial there)
// for matrices WITHOUT a grid: "chained" level scheduling.
//
// The level-scheduled kernels of bis_sptrsv.hip pay one hand-off through memory (store, poll: 2-3 us under load)
// per dependency LEVEL -- 1.7 thousand of them on the config-5 stand-in in its natural order, 6 thousand once an
// unstructured mesh has been RCM-ordered.  But the dependency that sets a row's level is, in a banded ordering,
// very often the row right before it (the previous unknown of the same node, the previous node of the same mesh
// line): on those inputs 85-95 % of the rows depend on their predecessor.  A CHAIN is a maximal run of consecutive
// rows (in substitution order) each of which depends on the one before; a PAIR OF WAVES solves a chain, row after row:
//
//   * inside the chain a result travels through registers / LDS (no trip to memory): the critical path of the
//     sweep counts a memory hand-off only where it crosses from one chain to another -- a few hundred to a
//     thousand times instead of once per level (tools/sweep_model.py);
//   * the consumer wave's lanes are the row's entries: acc = fma(val, x[col], acc) runs through the lanes in CRS
//     order (DPP), exactly the reference's arithmetic (bit-exact against the fma oracle);
//   * the feeder wave streams the chain's entries, b and D from memory, looks up the operands other chains
//     produce ("the data is the flag": one 8-byte sc1 store per row into a sentinel-initialised vector, polled
//     where it is not there yet) and hands value + operand to the consumer through an LDS ring, rows ahead of
//     it -- the consumer issues no load from memory at all, so it never waits for one (nor, through the shared
//     counter, for its own stores): a row that only waits for its predecessor costs LDS round trips and ALU.
//
// Chains are handed out by ticket in the order of the LEVEL OF THEIR FIRST ROW (eight ticket queues, a wave pair serves
// queue pair_id % 8), to a persistent grid whose waves are all resident; a pair holds one chain at a time.  This order is not a linear extension
// of the chain dependencies (chain A may need a row of a chain B that starts later), so progress needs an
// argument: let L be the lowest level with an unfinished row.  Such a row has all its operands; if its chain has
// been taken it is that chain's current row (rows of a chain have ascending levels) and runs.  Otherwise it is
// the FIRST row of an untaken chain, every taken chain of its queue starts at a level <= L and, if none of them
// can run, has its current row above L: they all straddle L.  The plan counts, per queue, the largest number of
// chains that straddle any level (first row's level <= L < last row's level) and the sweep is only used where
// that number is below the wave pairs serving the queue -- then a pair of the queue is free and takes the chain.
// Where the bound does not hold (levels wider than the machine: then there is parallelism to spare and latency
// is not the problem) the level-scheduled kernels run as before.  Every wait is bounded all the same and raises
// the context's fault word.
#include "bis_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <rocprim/rocprim.hpp>

struct bis_trsv_chain {
    int64_t n = 0;
    int n_chains = 0, n_queues = 0;
    int backward = 0;
    int32_t *c_row0 = nullptr;  // [n_chains] first row of chain t (ticket order: by the level of that row)
    int32_t *c_len = nullptr;   // [n_chains]
    unsigned long long *xs = nullptr; // [n + 1] published results, by row
    unsigned *ticket = nullptr;       // [n_queues * 64] one counter per queue, 256 bytes apart
    int max_straddle = 0;       // per queue, the largest number of chains that straddle a level
    double avg_len = 0.0;
};

void bis_trsv_chain_destroy(bis_trsv_chain *p) {
    if (!p) return;
    hipFree(p->c_row0); hipFree(p->c_len); hipFree(p->xs); hipFree(p->ticket);
    delete p;
}

namespace {

constexpr unsigned long long kSentinel = 0x7FF85EA71E55C0DEull; // quiet NaN + payload (same as bis_sptrsv.hip)
constexpr unsigned long long kCanonNaN = 0x7FF8000000000000ull;
constexpr unsigned kInternalTag = 0x7FF9C4A1u; // high word of an entry whose operand the chain produces itself (low word: how many rows back);
                                               // published results never carry it (a NaN result is canonicalised)
constexpr int kQueues = 8;
constexpr int kTicketStride = 64;     // unsigned words between the queues' counters
constexpr int kMaxChain = 128;        // rows per chain at most (= the consumer's ring of the chain's own results)
constexpr int kSlots = 8;             // entry ring: slots of 64 entries (one row, or one 64-entry segment of a longer row)
constexpr int kGroup = 4;             // slots the feeder loads together
constexpr int kRowBatch = 32;         // rows whose b / D / length the feeder fetches together
constexpr int kRowRing = 2 * kRowBatch;
constexpr unsigned kSpinMem = 1u << 20;  // polls of a memory word before a wave gives up (about a second)
constexpr unsigned kSpinLds = 1u << 26;  // polls of an LDS word (several seconds: longer than the feeder's own budget)
constexpr int kBlocksPerCU = 5;       // 23 KiB of LDS per workgroup; 4 per CU leaves each wave 128 VGPRs (with fewer the feeder, and at 7 per CU the consumer, spill: a trip to memory per row)

enum { C_GEN = 0, C_ROW0, C_M, C_WM, C_DONE, C_ROWS_DONE, C_CHAIN_DONE, C_EXIT, C_N = 8 };

// hand-offs between the two waves of a pair go through LDS words: the LDS executes one wave's operations in issue order, so
// "data, then the counter" needs no wait in between; all the code has to prevent is the compiler moving LDS accesses across
// the counter access (bis_trsv_tiled.hip has the long version of this note)
__device__ __forceinline__ unsigned lds_acquire(const unsigned *p) {
    const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void lds_release(unsigned *p, unsigned v) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ bool fault_raised(const unsigned *fault) {
    return __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
}

__global__ __launch_bounds__(256) void chain_fill_kernel(unsigned long long *xs, int64_t n, unsigned *ticket) {
    if (blockIdx.x == 0 && threadIdx.x < kQueues) ticket[threadIdx.x * kTicketStride] = 0u;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) xs[i] = kSentinel;
}

struct ChainArgs {
    const void *row_ptr;
    const int32_t *col;
    const double *val;
    const int32_t *c_row0, *c_len;
    const double *D, *b;
    double *x;
    unsigned long long *xs;
    unsigned *ticket;
    unsigned *fault;
    const int *stop;
    int64_t n;
    int n_chains;
    int idle_rounds; // poll rounds without a delivery after which a feeder polls one word only
    int idle_cap;    // ... and pauses up to this many x 256 cycles between its looks
    int prefix;      // 1: the feeder sums the entries before a row's first chain-internal operand
};

struct __attribute__((aligned(16))) Entry { double val; unsigned long long v; };

constexpr bool kFeederPrefix = false;     // the feeder sums the part of a row's fma chain before its first chain-internal operand: built, bit-exact, and
                                          // slower (the feeder then paces the pair: fem:80,80,81 forward 2.55 -> 3.0 ms) -- kept for the record, compiled out
constexpr int kSlotLen = 72;              // 64 entries + a group of (0.0, 0.0) entries behind them: what a fma chain reads past a slot's end
struct PairLds {
    Entry ent[kSlots][kSlotLen];          // the feeder's stream: value and operand of every entry, a slot per row segment
    double2 rbd[kRowRing];                // per row: b, D
    int rlen[kRowRing];                   // per row: number of entries
    unsigned long long res[kMaxChain];    // the chain's own results, by row & (kMaxChain - 1) (consumer only)
    double pre_acc[kSlots];               // per slot: the fma chain over the entries before the first chain-internal operand (feeder) ...
    int pre_q[kSlots];                    // ... and the index of that entry: where the consumer takes the chain up
    unsigned ctl[C_N];
};

// Two waves per chain.  The FEEDER (even wave) takes the ticket, streams the chain's entries and b / D from memory, looks up every
// operand another chain produces (polling those that are not there yet) and puts value + operand into the pair's LDS ring, a slot
// per row, in order; entries whose operand the chain itself produces carry a tag and the distance instead.  The CONSUMER (odd wave)
// issues NO loads from memory at all: it takes a row's entries from the ring, the chain's own results from registers (the last
// three) or its LDS ring, runs the fma chain across its lanes, divides and publishes.  A wave that both loads and stores waits for
// its own stores whenever it waits for a load (one counter, vmcnt, for both on gfx9): that wait -- a trip to memory per row -- is
// what the split removes; the consumer's row costs LDS round trips and ALU only.
template <typename RP, bool BACKWARD>
__global__ __launch_bounds__(256, kBlocksPerCU) void trsv_chain_kernel(const ChainArgs a) {
    __shared__ PairLds lds2[2];
    if (a.stop && a.stop[1]) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    PairLds &L = lds2[wv >> 1];
    const bool feeder = (wv & 1) == 0;
    if (threadIdx.x < 2 * C_N) lds2[threadIdx.x / C_N].ctl[threadIdx.x % C_N] = 0u;
    if (threadIdx.x < 2 * kSlots * 8) { // the padding group of every slot
        Entry &z = lds2[threadIdx.x / (kSlots * 8)].ent[(threadIdx.x / 8) % kSlots][64 + (threadIdx.x & 7)];
        z.val = 0.0; z.v = 0ull;
    }
    __syncthreads(); // (the only barrier: from here on the two pairs of the workgroup run on their own)
    const RP *rp = (const RP *)a.row_ptr;
    const unsigned pair_id = blockIdx.x * 2u + (unsigned)(wv >> 1);
    const int q = (int)(pair_id % (unsigned)kQueues);
    unsigned *ctr = a.ticket + q * kTicketStride;

    if (feeder) {
        unsigned gen = 0;
        unsigned slots_pub = 0; // slots published so far (all chains of this pair)
        for (;;) {
            unsigned k = 0;
            if (lane == 0) k = atomicAdd(ctr, 1u);
            k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
            const int64_t t = (int64_t)k * kQueues + q;
            if (t >= a.n_chains) { // this queue is empty: tell the consumer, leave
                if (lane == 0) lds_release(&L.ctl[C_EXIT], 1u);
                return;
            }
            const int row0 = __builtin_amdgcn_readfirstlane(a.c_row0[t]);
            const int m = __builtin_amdgcn_readfirstlane(a.c_len[t]);
            ++gen;
            if (lane == 0) {
                __hip_atomic_store(&L.ctl[C_ROW0], (unsigned)row0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&L.ctl[C_M], (unsigned)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&L.ctl[C_ROWS_DONE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                lds_release(&L.ctl[C_GEN], gen);
            }
            bool aborted = false;
            for (int base = 0; base < m; base += kRowBatch) {
                const int nb = min(kRowBatch, m - base);
                const int li = min(lane, nb - 1);
                const int my_r = BACKWARD ? row0 - (base + li) : row0 + (base + li);
                const int64_t my_s = (int64_t)rp[my_r];
                const int my_len = (int)((int64_t)rp[my_r + 1] - my_s);
                const double my_b = a.b[my_r], my_d = a.D[my_r];
                // the row ring holds two batches: this one may be written once the consumer is done with the one before the previous
                {
                    unsigned spins = 0;
                    while ((int)lds_acquire(&L.ctl[C_ROWS_DONE]) < base - kRowBatch) {
                        if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                if (lane < nb) {
                    L.rlen[(base + lane) & (kRowRing - 1)] = my_len;
                    L.rbd[(base + lane) & (kRowRing - 1)] = make_double2(my_b, my_d);
                }
                asm volatile("" ::: "memory");
                auto bcast64 = [&](long long v, int j) {
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, j);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), j);
                    return (long long)(((unsigned long long)hi << 32) | lo);
                };
                // segments of this batch in order: (row j, first entry, entries in the segment)
                int cur_j = 0, cur_off = 0; // next segment: row cur_j of the batch, entries from cur_off
                struct Seg { int j, cnt, first; int64_t k0; };
                auto next_seg = [&](Seg &sg) { // uniform; returns false when the batch is exhausted (the segment is then empty, addresses stay valid)
                    const bool have = cur_j < nb;
                    const int j = have ? cur_j : nb - 1;
                    const int len = __builtin_amdgcn_readlane(my_len, j);
                    const int64_t s = bcast64(my_s, j);
                    sg.j = j;
                    sg.k0 = s + (have ? cur_off : 0);
                    sg.cnt = have ? min(64, len - cur_off) : 0;
                    sg.first = cur_off == 0 ? 1 : 0;
                    if (have) { cur_off += 64; if (cur_off >= len) { ++cur_j; cur_off = 0; } }
                    return have;
                };
                struct Regs { int c; double av; };
                auto load_group = [&](Seg (&sg)[kGroup], Regs (&R)[kGroup]) { // UNCONDITIONAL loads from clamped addresses (no branch around a load:
                    int n_seg = 0;                                           // the compiler then counts its waits instead of draining everything)
#pragma unroll
                    for (int g = 0; g < kGroup; ++g) {
                        n_seg += next_seg(sg[g]) ? 1 : 0;
                        const int64_t ks = max(sg[g].k0 + min(lane, max(sg[g].cnt, 1) - 1), (int64_t)0);
                        R[g].c = a.col[ks];
                        R[g].av = a.val[ks];
                    }
                    return n_seg;
                };
                auto process_group = [&](const Seg (&sg)[kGroup], const Regs (&R)[kGroup], int n_seg) {
                    unsigned long long v[kGroup], raw[kGroup];
                    bool ext[kGroup], internal[kGroup];
                    int dist[kGroup];
#pragma unroll
                    for (int g = 0; g < kGroup; ++g) { // first look at every operand, all of the group in flight together (lanes without an entry /
                                                       // with a chain-internal operand look at a word nobody needs: no branch around the load)
                        const int r = BACKWARD ? row0 - (base + sg[g].j) : row0 + (base + sg[g].j);
                        dist[g] = BACKWARD ? R[g].c - r : r - R[g].c;
                        const bool act = lane < sg[g].cnt;
                        internal[g] = act && dist[g] > 0 && dist[g] <= base + sg[g].j;
                        ext[g] = act && !internal[g];
                        raw[g] = __hip_atomic_load(&a.xs[ext[g] ? R[g].c : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    int qarr[kGroup]; // per slot: the first entry with a chain-internal operand (the row's length if there is none; 0 for the
                                      // later segments of a long row, whose chain continues the previous segment's)
#pragma unroll
                    for (int g = 0; g < kGroup; ++g) {
                        v[g] = internal[g] ? (((unsigned long long)kInternalTag << 32) | (unsigned)dist[g]) : ext[g] ? raw[g] : 0ull;
                        const unsigned long long im = __ballot(internal[g]);
                        qarr[g] = (kFeederPrefix && sg[g].first && a.prefix) ? (im ? (int)__builtin_ctzll(im) : sg[g].cnt) : 0;
                    }
                    // ring space: the slots this group takes must have been consumed
                    {
                        unsigned spins = 0;
                        while ((int)(slots_pub + (unsigned)n_seg - lds_acquire(&L.ctl[C_DONE])) > kSlots) {
                            if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < kGroup; ++g) // values first: the operands follow as they arrive
                        if (g < n_seg) L.ent[(slots_pub + (unsigned)g) & (kSlots - 1)][lane].val = lane < sg[g].cnt ? R[g].av : 0.0;
                    // Slots are published IN ORDER, each when all its operands are there (an operand of row j can only depend on rows of
                    // this chain before j, whose slots are out already: waiting here never waits for the consumer's own future).  All
                    // pending operands of the group are polled together, so operands that neighbouring chains produce row by row are
                    // picked up a group per round trip, not a row per round trip.  A feeder whose rounds deliver nothing -- a chain taken
                    // long before its turn: most of the resident pairs at any time -- falls back to ONE lane polling ONE word (the operand
                    // produced last: the largest column of a forward, the smallest of a backward sweep) with a growing pause: thousands
                    // of idle feeders re-reading all their pending words were most of the load on the L2 the active chains' hand-offs
                    // queue behind.
                    int pub = 0, idle = 0;
                    bool delivered = false;
                    unsigned spins = 0;
                    for (;;) {
                        bool progressed = false;
                        int pub1 = pub;
#pragma unroll
                        for (int g = 0; g < kGroup; ++g) {
                            if (g == pub1 && g < n_seg && !__ballot(ext[g] && v[g] == kSentinel)) {
                                L.ent[(slots_pub + (unsigned)g) & (kSlots - 1)][lane].v = v[g];
                                ++pub1;
                            }
                        }
                        if (pub1 > pub) {
                            // The part of each row's fma chain that needs nothing of this chain -- the entries before its first chain-internal
                            // operand: in a forward sweep over ascending columns all but the last few -- is summed HERE, off the consumer's
                            // critical path: lane j takes the j-th slot of this round, the slots' chains run side by side (same CRS order,
                            // same fma per entry: the consumer continues where this stops).
                            int myq = 0, maxq = 0;
                            unsigned myslot = 0;
#pragma unroll
                            for (int g = 0; g < kGroup; ++g) {
                                if (g >= pub && g < pub1) {
                                    maxq = max(maxq, qarr[g]);
                                    if (lane == g - pub) { myq = qarr[g]; myslot = (slots_pub + (unsigned)g) & (kSlots - 1); }
                                }
                            }
                            asm volatile("" ::: "memory");
                            if (kFeederPrefix && lane < pub1 - pub) {
                                double acc = 0.0;
                                typedef double v2d_lds __attribute__((ext_vector_type(2)));
                                const volatile __attribute__((address_space(3))) v2d_lds *q =
                                    (const volatile __attribute__((address_space(3))) v2d_lds *)&L.ent[myslot][0];
                                for (int i0 = 0; i0 < maxq; i0 += 4) {
                                    v2d_lds e[4];
#pragma unroll
                                    for (int u = 0; u < 4; ++u) e[u] = q[i0 + u];
#pragma unroll
                                    for (int u = 0; u < 4; ++u) {
                                        const double t = fma(e[u].x, e[u].y, acc);
                                        acc = i0 + u < myq ? t : acc;
                                    }
                                }
                                L.pre_acc[myslot] = acc;
                                L.pre_q[myslot] = myq;
                            }
                            if (lane == 0) lds_release(&L.ctl[C_WM], slots_pub + (unsigned)pub1);
                            pub = pub1;
                            progressed = true;
                        }
                        if (pub >= n_seg) break;
                        idle = (progressed || delivered) ? 0 : idle + 1; // (a delivery that does not complete a slot yet still says the wavefront is here)
                        delivered = false;
                        ++spins;
                        if (!aborted && (spins & 255u) == 0u) aborted = __builtin_amdgcn_readfirstlane((int)fault_raised(a.fault)) != 0;
                        if (aborted || spins > kSpinMem) { // bounded: the rows get NaN, the context's fault word is raised
                            aborted = true;
                            if (lane == 0) __hip_atomic_fetch_or(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
                            for (int g = 0; g < kGroup; ++g) if (ext[g] && v[g] == kSentinel) v[g] = kCanonNaN;
                            continue;
                        }
                        if (idle < a.idle_rounds) {
#pragma unroll
                            for (int g = 0; g < kGroup; ++g)
                                if (g >= pub && g < n_seg && ext[g] && v[g] == kSentinel)
                                    v[g] = __hip_atomic_load(&a.xs[R[g].c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __builtin_amdgcn_s_sleep(1);
                            // (deliveries show as published slots in the next round)
                        } else {
#pragma unroll
                            for (int g = 0; g < kGroup; ++g) {
                                if (g == pub) {
                                    const unsigned long long pend = __ballot(ext[g] && v[g] == kSentinel);
                                    const int sel = BACKWARD ? (int)__builtin_ctzll(pend) : 63 - (int)__builtin_clzll(pend);
                                    if (lane == sel) v[g] = __hip_atomic_load(&a.xs[R[g].c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    delivered = __ballot(lane == sel && v[g] != kSentinel) != 0ull;
                                }
                            }
                            const int pause = min(idle - a.idle_rounds + 1, a.idle_cap);
                            for (int i = 0; i < pause; ++i) __builtin_amdgcn_s_sleep(4);
                        }
                    }
                    slots_pub += (unsigned)n_seg;
                };
                Seg sA[kGroup], sB[kGroup];
                Regs rA[kGroup], rB[kGroup];
                int nA = load_group(sA, rA);
                for (;;) { // (two register sets, roles swapped by unrolling: the next group's entries are in flight while this group's operands are looked up)
                    const int nB = load_group(sB, rB);
                    process_group(sA, rA, nA);
                    if (nB == 0) break;
                    nA = load_group(sA, rA);
                    process_group(sB, rB, nB);
                    if (nA == 0) break;
                }
            }
            // the pair holds ONE chain at a time (the progress argument counts chains that are being worked on): wait for the consumer
            {
                unsigned spins = 0;
                while (lds_acquire(&L.ctl[C_CHAIN_DONE]) != gen) {
                    if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
        }
    } else {
        __builtin_amdgcn_s_setprio(3); // the row chain is the sweep's critical path: issue before the feeders that share the SIMD
        unsigned gen = 0;
        unsigned slot = 0; // next slot to consume
        for (;;) {
            ++gen;
            {
                unsigned spins = 0;
                for (;;) {
                    if (lds_acquire(&L.ctl[C_GEN]) == gen) break;
                    if (lds_acquire(&L.ctl[C_EXIT])) return;
                    if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            const int row0 = (int)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&L.ctl[C_ROW0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            const int m = (int)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&L.ctl[C_M], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            unsigned long long p1 = 0ull, p2 = 0ull, p3 = 0ull; // the chain's last three results
            unsigned wm = 0; // cached watermark
            auto wait_slot = [&](unsigned need) { // slots below `need` are published
                if ((int)(wm - need) >= 0) return;
                unsigned spins = 0;
                for (;;) {
                    wm = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_acquire(&L.ctl[C_WM]));
                    if ((int)(wm - need) >= 0) break;
                    if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            };
            // (the next row's length, b / D, prefix and first entries are read while this row is divided and published: used if they were there)
            bool have_n = false;
            int len_n = 0, q_n = 0;
            double2 bd_n = make_double2(0.0, 1.0);
            double acc_n = 0.0;
            unsigned long long ev_n = 0ull;
            for (int j = 0; j < m; ++j) {
                const int r = BACKWARD ? row0 - j : row0 + j;
                // the row's length, b / D, the feeder's prefix of its fma chain and this lane's operand of its first slot are read TOGETHER with
                // the watermark that says whether they are there (the LDS runs a wave's reads in order: a watermark that covers the slot was
                // read before data that is then valid) -- one LDS round trip per row
                int len = len_n, q0 = q_n;
                double2 bd = bd_n;
                double acc = acc_n;
                unsigned long long ev = ev_n;
                if (!have_n) {
                    unsigned spins = 0;
                    for (;;) {
                        wm = lds_acquire(&L.ctl[C_WM]);
                        len = L.rlen[j & (kRowRing - 1)];
                        bd = L.rbd[j & (kRowRing - 1)];
                        if (kFeederPrefix) { acc = L.pre_acc[slot & (kSlots - 1)]; q0 = L.pre_q[slot & (kSlots - 1)]; }
                        else { acc = 0.0; q0 = 0; }
                        ev = L.ent[slot & (kSlots - 1)][lane].v;
                        asm volatile("" ::: "memory");
                        wm = (unsigned)__builtin_amdgcn_readfirstlane((int)wm);
                        if ((int)(wm - (slot + 1u)) >= 0) break;
                        if (++spins > kSpinLds) { if (lane == 0) __hip_atomic_fetch_or(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                len = __builtin_amdgcn_readfirstlane(len);
                q0 = __builtin_amdgcn_readfirstlane(q0);
                int done = 0;
                do { // a slot per 64 entries (at least one per row)
                    Entry (&S)[kSlotLen] = L.ent[slot & (kSlots - 1)];
                    if (done > 0) { wait_slot(slot + 1u); ev = S[lane].v; q0 = 0; }
                    const int cnt = min(64, len - done);
                    // Operands this chain produced itself carry the feeder's tag and how many rows back they were made: the last three are in
                    // registers (p1..p3), older ones in the ring of the chain's results; the lanes that hold them write them into the slot.
                    // (Picking them inside the one-lane chain instead -- a compare and a few selects per entry -- made every entry cost 64 ns:
                    // a lone lane's instruction issue, not the fma's latency, then paces the chain.)
                    // (branch-free selects and one store under the tag: as nested ?: the compiler made it a chain of branches, 1.4x the cycles)
                    {
                        const bool tagged = (unsigned)(ev >> 32) == kInternalTag;
                        const int dist = (int)(unsigned)ev;
                        unsigned long long w = p3;
                        w = dist == 2 ? p2 : w;
                        w = dist == 1 ? p1 : w;
                        if (__ballot(tagged && dist > 3)) {
                            const int orow = BACKWARD ? r + dist : r - dist;
                            const unsigned long long o = __hip_atomic_load(&L.res[orow & (kMaxChain - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            w = dist > 3 ? o : w;
                        }
                        if (tagged) __hip_atomic_store(&S[lane].v, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    asm volatile("" ::: "memory");
                    // acc = fma(a_i, x_i, acc) in CRS order -- the reference's chain exactly -- by ONE lane, from entry q0 on (the feeder has
                    // summed the entries before it), reading entry after entry from the slot: four entries in flight ahead of the four being
                    // summed (a one-lane LDS read moves 16 bytes; the chain costs the dependent fma's latency, ~10 ns per entry: with DPP moves
                    // across the lanes it measured 32 ns, with every lane reading every entry 12.5 ns -- tools/chain_probe.py).  The slot is
                    // padded with (0.0, 0.0) entries: fma(0, 0, acc) leaves acc as it is (acc is never -0.0: it starts at +0.0 and an exact
                    // cancellation rounds to +0.0), so the count is rounded up to the group and no step is predicated.
                    if (lane == 0) {
                        typedef double v2d_lds __attribute__((ext_vector_type(2)));
                        // (volatile so that the reads stay where they are written -- an ordinary read whose value is only used if the loop goes
                        // on is sunk behind the loop's exit test --, in the LDS address space so that they stay LDS reads)
                        const volatile __attribute__((address_space(3))) v2d_lds *q = (const volatile __attribute__((address_space(3))) v2d_lds *)&S[0];
                        // (two register sets of eight entries, roles swapped by unrolling: eight fmas -- ~170 cycles of dependent latency -- cover
                        // the LDS latency of the next eight reads; with sets of four the chain stalled on its reads)
                        v2d_lds A8[8], B8[8];
                        const int qs = kFeederPrefix ? q0 : 0;
#pragma unroll
                        for (int u = 0; u < 8; ++u) A8[u] = q[qs + u];
                        for (int e0 = qs; e0 < cnt; e0 += 16) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) B8[u] = q[min(e0 + 8, 64) + u];
#pragma unroll
                            for (int u = 0; u < 8; ++u) acc = fma(A8[u].x, A8[u].y, acc);
#pragma unroll
                            for (int u = 0; u < 8; ++u) A8[u] = q[min(e0 + 16, 64) + u];
                            if (e0 + 8 < cnt) {
#pragma unroll
                                for (int u = 0; u < 8; ++u) acc = fma(B8[u].x, B8[u].y, acc);
                            }
                        }
                    }
                    done += 64;
                    ++slot;
                    // a long row gives its slots back segment by segment (the sum so far is in a register): the feeder's space check
                    // counts slots, and a row of more segments than the ring has slots -- or one whose last segment falls into the
                    // feeder's next group -- would otherwise wait for a release that only its own last segment brings.  (The last
                    // segment's release stays behind the publish below: rows of <= 64 entries run as before.)
                    if (done < len && lane == 0) lds_release(&L.ctl[C_DONE], slot);
                } while (done < len);
                {   // optimistic look at the next row (valid if the watermark, read first, covers its slot)
                    const unsigned wmn = lds_acquire(&L.ctl[C_WM]);
                    len_n = L.rlen[(j + 1) & (kRowRing - 1)];
                    bd_n = L.rbd[(j + 1) & (kRowRing - 1)];
                    if (kFeederPrefix) { acc_n = L.pre_acc[slot & (kSlots - 1)]; q_n = L.pre_q[slot & (kSlots - 1)]; }
                    else { acc_n = 0.0; q_n = 0; }
                    ev_n = L.ent[slot & (kSlots - 1)][lane].v;
                    asm volatile("" ::: "memory");
                    wm = wmn;
                }
                unsigned long long out = 0ull;
                if (lane == 0) {
                    const double res = (bd.x - acc) / bd.y;
                    out = (unsigned long long)__double_as_longlong(res);
                    if (res != res) out = kCanonNaN; // never publish the sentinel (or the tag) pattern
                    __hip_atomic_store(&a.xs[r], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // sc1: the flag IS the data
                    a.x[r] = __longlong_as_double((long long)out);
                    __hip_atomic_store(&L.res[r & (kMaxChain - 1)], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    lds_release(&L.ctl[C_DONE], slot);
                    lds_release(&L.ctl[C_ROWS_DONE], (unsigned)(j + 1));
                }
                out = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(out >> 32)) << 32) |
                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)out);
                p3 = p2; p2 = p1; p1 = out;
                wm = (unsigned)__builtin_amdgcn_readfirstlane((int)wm);
                have_n = j + 1 < m && (int)(wm - (slot + 1u)) >= 0;
            }
            if (lane == 0) lds_release(&L.ctl[C_CHAIN_DONE], gen);
        }
    }
}

// ---- plan --------------------------------------------------------------------------------------------------------

// linked[p] = 1 iff the row at position p of the substitution order has the row at position p - 1 among its operands
template <typename RP, bool BACKWARD>
__global__ __launch_bounds__(256) void chain_link_kernel(const RP *__restrict__ rp, const int32_t *__restrict__ col, int64_t n,
                                                         int32_t *__restrict__ head0) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += stride) {
        const int64_t r = BACKWARD ? n - 1 - p : p;
        const int32_t pred = (int32_t)(BACKWARD ? r + 1 : r - 1);
        bool linked = false;
        if (p > 0)
            for (int64_t k = rp[r]; k < rp[r + 1]; ++k) linked |= col[k] == pred;
        head0[p] = linked ? 0 : 1;
    }
}

// run_start[id] = position of the run's first row (id = inclusive scan of head0, minus one)
__global__ __launch_bounds__(256) void chain_run_start_kernel(const int32_t *__restrict__ head0, const int32_t *__restrict__ run_id, int64_t n,
                                                              int32_t *__restrict__ run_start) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += stride)
        if (head0[p]) run_start[run_id[p] - 1] = (int32_t)p;
}

// runs are cut every kMaxChain rows
__global__ __launch_bounds__(256) void chain_cut_kernel(const int32_t *__restrict__ head0, const int32_t *__restrict__ run_id,
                                                        const int32_t *__restrict__ run_start, int64_t n, int32_t *__restrict__ head) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += stride)
        head[p] = (head0[p] || ((p - run_start[run_id[p] - 1]) % kMaxChain) == 0) ? 1 : 0;
}

// per chain (unsorted, by position): first position, and the levels of its first and last row
template <bool BACKWARD>
__global__ __launch_bounds__(256) void chain_collect_kernel(const int32_t *__restrict__ head, const int32_t *__restrict__ chain_id, int64_t n,
                                                            const int *__restrict__ level, int32_t *__restrict__ pos0, int *__restrict__ lvl_first,
                                                            int *__restrict__ lvl_last, int32_t *__restrict__ iota) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += stride) {
        const int64_t r = BACKWARD ? n - 1 - p : p;
        const int id = chain_id[p] - 1;
        if (head[p]) { pos0[id] = (int32_t)p; lvl_first[id] = level[r]; iota[id] = id; }
        if (p == n - 1 || head[p + 1]) lvl_last[id] = level[r];
    }
}

template <bool BACKWARD>
__global__ __launch_bounds__(256) void chain_emit_kernel(const int32_t *__restrict__ order, const int32_t *__restrict__ pos0, int64_t n, int n_chains,
                                                         const int *__restrict__ lvl_last, int32_t *__restrict__ c_row0, int32_t *__restrict__ c_len,
                                                         int *__restrict__ last_sorted) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_chains) return;
    const int id = order[t];
    const int64_t p = pos0[id];
    const int64_t pe = id + 1 < n_chains ? (int64_t)pos0[id + 1] : n;
    c_row0[t] = (int32_t)(BACKWARD ? n - 1 - p : p);
    c_len[t] = (int32_t)(pe - p);
    last_sorted[t] = lvl_last[id];
}

struct ChainBufs {
    std::vector<void *> v;
    hipError_t e = hipSuccess;
    template <class T> T *get(size_t count) {
        void *p = nullptr;
        if (e == hipSuccess) e = hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16));
        if (e == hipSuccess) v.push_back(p); else p = nullptr;
        return (T *)p;
    }
    ~ChainBufs() { for (void *p : v) hipFree(p); }
};

int resident_blocks(bool rp64, bool backward) {
    static int res[4] = {0, 0, 0, 0};
    int &r = res[(rp64 ? 2 : 0) + (backward ? 1 : 0)];
    if (r == 0) {
        int nb = 0;
        hipError_t oe;
        if (rp64) oe = backward ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trsv_chain_kernel<int64_t, true>, 256, 0)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trsv_chain_kernel<int64_t, false>, 256, 0);
        else oe = backward ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trsv_chain_kernel<int32_t, true>, 256, 0)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trsv_chain_kernel<int32_t, false>, 256, 0);
        r = (oe == hipSuccess && nb > 0) ? std::min(nb, kBlocksPerCU) : 1;
        (void)hipGetLastError();
    }
    return r;
}

int chain_grid(const bis_ctx *ctx, bool rp64, bool backward) {
    const int share = std::max(1, bis_opts().device_share);
    int g = std::max(4, ctx->n_cus * resident_blocks(rp64, backward) / share);
    return g & ~3; // 2 wave pairs per workgroup: a grid that is a multiple of 4 serves each of the 8 queues with the same number of pairs
}

} // namespace

int bis_trsv_chain_resident_pairs(const bis_ctx *ctx, bool rp64, bool backward) { return chain_grid(ctx, rp64, backward) * 2; }

// *out stays null (BIS_OK) where the chained sweep does not apply: short chains, or more chains straddling a level than
// the resident waves can hold (see the header).  level_dev: the dependency levels of T's rows (bis_trsv_analyse_device).
bis_status bis_trsv_chain_build(bis_ctx *ctx, const bis_mat *T, bool backward, const int *level_dev, int n_levels,
                                bis_trsv_chain **out) {
    *out = nullptr;
    const int64_t n = T->n_rows;
    if (n < 2 || T->nnz == 0 || T->view || n >= INT32_MAX || !level_dev) return BIS_OK;
    hipStream_t s = ctx->stream;
    ChainBufs B;
    auto fail = [&](hipError_t e) {
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return BIS_OK; } // an optimisation that does not fit: not an error
        ctx->err = std::string("chained sptrsv plan: ") + hipGetErrorString(e);
        return BIS_ERR_HIP;
    };
#define CH_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_); } while (0)
    const int grid_n = (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->n_cus * 16);
    int32_t *head0 = B.get<int32_t>((size_t)n), *run_id = B.get<int32_t>((size_t)n), *head = B.get<int32_t>((size_t)n), *chain_id = B.get<int32_t>((size_t)n);
    CH_CHECK(B.e);
#define CH_LAUNCH(K, ...) do { \
        if (T->rp64) { if (backward) hipLaunchKernelGGL((K<int64_t, true>), dim3(grid_n), dim3(256), 0, s, (const int64_t *)T->row_ptr, __VA_ARGS__); \
                       else hipLaunchKernelGGL((K<int64_t, false>), dim3(grid_n), dim3(256), 0, s, (const int64_t *)T->row_ptr, __VA_ARGS__); } \
        else { if (backward) hipLaunchKernelGGL((K<int32_t, true>), dim3(grid_n), dim3(256), 0, s, (const int32_t *)T->row_ptr, __VA_ARGS__); \
               else hipLaunchKernelGGL((K<int32_t, false>), dim3(grid_n), dim3(256), 0, s, (const int32_t *)T->row_ptr, __VA_ARGS__); } \
        CH_CHECK(hipGetLastError()); } while (0)
    CH_LAUNCH(chain_link_kernel, T->col, n, head0);
    size_t b_scan = 0;
    CH_CHECK(rocprim::inclusive_scan(nullptr, b_scan, head0, run_id, (size_t)n, rocprim::plus<int32_t>(), s));
    char *tmp = B.get<char>(b_scan);
    CH_CHECK(B.e);
    size_t bb = b_scan;
    CH_CHECK(rocprim::inclusive_scan(tmp, bb, head0, run_id, (size_t)n, rocprim::plus<int32_t>(), s));
    int32_t n_runs = 0;
    CH_CHECK(hipMemcpyAsync(&n_runs, run_id + (n - 1), 4, hipMemcpyDeviceToHost, s));
    CH_CHECK(hipStreamSynchronize(s));
    int32_t *run_start = B.get<int32_t>((size_t)n_runs);
    CH_CHECK(B.e);
    hipLaunchKernelGGL(chain_run_start_kernel, dim3(grid_n), dim3(256), 0, s, head0, run_id, n, run_start);
    hipLaunchKernelGGL(chain_cut_kernel, dim3(grid_n), dim3(256), 0, s, head0, run_id, run_start, n, head);
    CH_CHECK(hipGetLastError());
    bb = b_scan;
    CH_CHECK(rocprim::inclusive_scan(tmp, bb, head, chain_id, (size_t)n, rocprim::plus<int32_t>(), s));
    int32_t n_chains = 0;
    CH_CHECK(hipMemcpyAsync(&n_chains, chain_id + (n - 1), 4, hipMemcpyDeviceToHost, s));
    CH_CHECK(hipStreamSynchronize(s));
    const double avg_len = (double)n / (double)std::max(n_chains, 1);
    const bool stats = getenv("BIS_TRSV_CHAIN_STATS") != nullptr;
    const int want = bis_opts().trsv_chain;
    if (want < 1 && avg_len < 3.0) { // mostly single rows: one ticket per row buys nothing over the level-scheduled kernels
        if (stats) fprintf(stderr, "chained sptrsv plan (%s): %lld rows in %d chains (%.2f rows each): too short, not used\n", backward ? "backward" : "forward", (long long)n, n_chains, avg_len);
        return BIS_OK;
    }
    int32_t *pos0 = B.get<int32_t>((size_t)n_chains), *iota = B.get<int32_t>((size_t)n_chains), *order = B.get<int32_t>((size_t)n_chains);
    int *lvl_first = B.get<int>((size_t)n_chains), *lvl_last = B.get<int>((size_t)n_chains), *lvl_sorted = B.get<int>((size_t)n_chains),
        *last_sorted = B.get<int>((size_t)n_chains);
    CH_CHECK(B.e);
    if (backward) hipLaunchKernelGGL(chain_collect_kernel<true>, dim3(grid_n), dim3(256), 0, s, head, chain_id, n, level_dev, pos0, lvl_first, lvl_last, iota);
    else hipLaunchKernelGGL(chain_collect_kernel<false>, dim3(grid_n), dim3(256), 0, s, head, chain_id, n, level_dev, pos0, lvl_first, lvl_last, iota);
    CH_CHECK(hipGetLastError());
    int bits = 1;
    while ((1ll << bits) < std::max(n_levels, 2)) ++bits;
    size_t b_sort = 0;
    CH_CHECK(rocprim::radix_sort_pairs(nullptr, b_sort, lvl_first, lvl_sorted, iota, order, (size_t)n_chains, 0, bits, s));
    char *tmp2 = B.get<char>(b_sort);
    CH_CHECK(B.e);
    CH_CHECK(rocprim::radix_sort_pairs(tmp2, b_sort, lvl_first, lvl_sorted, iota, order, (size_t)n_chains, 0, bits, s)); // stable: chains of one level in substitution order
    bis_trsv_chain *p = new bis_trsv_chain;
    struct Guard { bis_trsv_chain *p; ~Guard() { bis_trsv_chain_destroy(p); } } guard{p};
    p->n = n; p->n_chains = n_chains; p->n_queues = kQueues; p->backward = backward ? 1 : 0; p->avg_len = avg_len;
    CH_CHECK(hipMalloc(&p->c_row0, 4 * (size_t)n_chains));
    CH_CHECK(hipMalloc(&p->c_len, 4 * (size_t)n_chains));
    CH_CHECK(hipMalloc(&p->xs, 8 * (size_t)(n + 1)));
    CH_CHECK(hipMalloc(&p->ticket, sizeof(unsigned) * kQueues * kTicketStride));
    const unsigned gc = (unsigned)((n_chains + 255) / 256);
    if (backward) hipLaunchKernelGGL(chain_emit_kernel<true>, dim3(gc), dim3(256), 0, s, order, pos0, n, n_chains, lvl_last, p->c_row0, p->c_len, last_sorted);
    else hipLaunchKernelGGL(chain_emit_kernel<false>, dim3(gc), dim3(256), 0, s, order, pos0, n, n_chains, lvl_last, p->c_row0, p->c_len, last_sorted);
    CH_CHECK(hipGetLastError());
    // the residency bound, per queue: chains that straddle a level (first row's level <= L < last row's level)
    std::vector<int> hf((size_t)n_chains), hl((size_t)n_chains);
    CH_CHECK(hipMemcpyAsync(hf.data(), lvl_sorted, 4 * (size_t)n_chains, hipMemcpyDeviceToHost, s));
    CH_CHECK(hipMemcpyAsync(hl.data(), last_sorted, 4 * (size_t)n_chains, hipMemcpyDeviceToHost, s));
    CH_CHECK(hipStreamSynchronize(s));
#undef CH_LAUNCH
#undef CH_CHECK
    int worst = 0;
    {
        std::vector<int> diff((size_t)n_levels + 2);
        for (int qq = 0; qq < kQueues; ++qq) {
            std::fill(diff.begin(), diff.end(), 0);
            for (int64_t t = qq; t < n_chains; t += kQueues)
                if (hl[(size_t)t] > hf[(size_t)t]) { diff[(size_t)hf[(size_t)t]] += 1; diff[(size_t)hl[(size_t)t]] -= 1; }
            int run = 0;
            for (int l = 0; l <= n_levels; ++l) { run += diff[(size_t)l]; worst = std::max(worst, run); }
        }
    }
    p->max_straddle = worst;
    const int waves_per_queue = chain_grid(ctx, T->rp64, backward) * 2 / kQueues; // (wave PAIRS: one chain each)
    const bool fits = worst + 2 <= waves_per_queue;
    if (stats)
        fprintf(stderr, "chained sptrsv plan (%s): %lld rows, %d levels, %d chains (%.2f rows each), at most %d chains of a queue straddle a level, "
                        "%d wave pairs per queue: %s\n", backward ? "backward" : "forward", (long long)n, n_levels, n_chains, avg_len, worst, waves_per_queue,
                fits ? "used" : "NOT used (the resident waves could not hold them)");
    if (!fits) return BIS_OK;
    guard.p = nullptr;
    *out = p;
    return BIS_OK;
}

bis_status bis_trsv_chain_solve(bis_ctx *ctx, const bis_mat *T, bis_trsv_chain *p, double *x, const double *D, const double *b) {
    const int fill_grid = (int)std::min<int64_t>((p->n + 1 + 255) / 256, 2048);
    hipLaunchKernelGGL(chain_fill_kernel, dim3(fill_grid), dim3(256), 0, ctx->stream, p->xs, p->n + 1, p->ticket);
    ChainArgs a{T->row_ptr, T->col, T->val, p->c_row0, p->c_len, D, b, x, p->xs, p->ticket, ctx->fault_dev, ctx->spmv_stop, p->n, p->n_chains,
                bis_opts().trsv_chain_idle >= 0 ? bis_opts().trsv_chain_idle : (1 << 30), bis_opts().trsv_chain_pause >= 0 ? bis_opts().trsv_chain_pause : 8,
                bis_opts().trsv_chain_prefix > 0 ? 1 : 0};
    // every wave of the grid must be resident (the progress argument counts them): the grid is what the occupancy query allows
    int grid = chain_grid(ctx, T->rp64, p->backward != 0);
    // fewer pairs = fewer feeders waiting far ahead of the wavefront; never fewer than the progress argument needs (per queue: the
    // chains that may straddle a level, plus one that is free)
    // (measured, pairs per queue -> ms per sweep: fem:40,40,41 [60 straddle] 128: 0.92, 320: 1.12; fem:80,80,81 [229] 256: 2.44, 320: 2.62;
    // unstr:80,80,80 RCM-ordered [88 forward / 220 backward] 128 (222 backward): 6.6 / 7.0, 256: 7.2 / 7.0, 320: 7.7 / 7.5)
    // With few pairs beyond the bound, and every feeder polling all its pending operands all the time (no idle mode):
    // fem:80,80,81 231 pairs: 2.05 ms (320 pairs with the idle mode 2.62); unstr:80,80,80 RCM 90 / 222 pairs: 6.35 / 7.33 ms (6.9 / 7.7).
    const int pairs = bis_opts().trsv_chain_pairs > 0 ? bis_opts().trsv_chain_pairs : p->max_straddle + p->max_straddle / 8 + 8;
    grid = std::max(std::min(grid, (pairs * kQueues / 2 + 3) & ~3), std::min(grid, ((p->max_straddle + 2) * kQueues / 2 + 3) & ~3));
    if (T->rp64) {
        if (p->backward) hipLaunchKernelGGL((trsv_chain_kernel<int64_t, true>), dim3(grid), dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL((trsv_chain_kernel<int64_t, false>), dim3(grid), dim3(256), 0, ctx->stream, a);
    } else {
        if (p->backward) hipLaunchKernelGGL((trsv_chain_kernel<int32_t, true>), dim3(grid), dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL((trsv_chain_kernel<int32_t, false>), dim3(grid), dim3(256), 0, ctx->stream, a);
    }
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}
