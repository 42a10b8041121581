// handoff_bench.hip -- latency of a producer -> consumer hand-off through memory between two workgroups, by where
// the two run (same XCD / another XCD) and by how the word is published.  Decides whether the tiled sweep's
// hand-offs (bis_trsv_tiled.hip) gain from keeping a dependency chain inside one XCD.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/handoff_bench tools/handoff_bench.hip && gpurun_out/handoff_bench
// Workgroup 0 plays ping against every other workgroup in turn: kTrips round trips of one 8-byte word each way.
//   mode 0: agent-scope store (sc1: written through to memory), agent-scope load -- what the sweep does today
//   mode 1: workgroup-scope store (stays in the producer's L2), agent-scope load (L1 bypassed, L2 served)
//   mode 2: as 1, the consumer's 64 lanes polling 49 different lines per round (the poller's situation)
//   mode 3: as 0 (sc1 stores), polled as in mode 2: are agent-scope loads of lines nobody writes served by the L2?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kTrips = 400;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__device__ __forceinline__ void put(unsigned long long *p, unsigned long long v) {
    if (MODE == 0 || MODE == 3) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned long long get(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// words: [partner][2][16] (a 128-byte line per direction)
template <int MODE>
__global__ __launch_bounds__(64) void pingpong(unsigned long long *words, unsigned *turn, unsigned *xcc, long long *ticks, int *lost, long long *rounds) {
    const int wg = blockIdx.x, n = gridDim.x;
    if (threadIdx.x == 0) xcc[wg] = xcc_id();
    if (wg == 0) {
        for (int p = 1; p < n; ++p) {
            if ((MODE == 1 || MODE == 2) && (p & 7) != 0) continue; // (workgroups go to the XCDs round robin: a store kept in L2 only reaches the same XCD)
            unsigned long long *to = words + (size_t)p * 32, *from = to + 16;
            if (threadIdx.x == 0) __hip_atomic_store(turn, (unsigned)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            for (int k = 1; k <= kTrips; ++k) {
                if (threadIdx.x == 0) put<MODE>(to, (unsigned long long)k);
                unsigned spins = 0;
                while (get(from) != (unsigned long long)k) {
                    if (++spins > (1u << 22)) { if (threadIdx.x == 0) *lost = p; return; }
                }
            }
            if (threadIdx.x == 0) ticks[p] = (long long)__builtin_amdgcn_s_memrealtime() - t0;
        }
        if (threadIdx.x == 0) __hip_atomic_store(turn, (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        unsigned long long *from = words + (size_t)wg * 32, *to = from + 16;
        unsigned spins = 0;
        for (;;) { // wait for the turn (or for the end, if ping gave up)
            const unsigned t = __hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (unsigned)wg) break;
            if (t > (unsigned)wg || ((MODE == 1 || MODE == 2) && (wg & 7) != 0) || ++spins > (1u << 24) || __hip_atomic_load(lost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
            __builtin_amdgcn_s_sleep(32);
        }
        long long n_rounds = 0;
        const long long c0 = (long long)__builtin_readcyclecounter();
        for (int k = 1; k <= kTrips; ++k) {
            spins = 0;
            if (MODE >= 2) {
                // the lanes poll 16 different words of other lines too, as a poller with many operands would
                for (;;) {
                    const unsigned long long v = get(threadIdx.x < 16 ? from : words + (size_t)((wg + threadIdx.x) % gridDim.x) * 32);
                    const unsigned long long v0 = __shfl(v, 0, 64);
                    ++n_rounds;
                    if (v0 == (unsigned long long)k) break;
                    if (++spins > (1u << 22)) return;
                }
            } else {
                for (;;) { ++n_rounds; if (get(from) == (unsigned long long)k) break; if (++spins > (1u << 22)) return; }
            }
            if (threadIdx.x == 0) put<MODE>(to, (unsigned long long)k);
        }
        if (threadIdx.x == 0) { rounds[2 * wg] = n_rounds; rounds[2 * wg + 1] = (long long)__builtin_readcyclecounter() - c0; }
    }
}

int main() {
    const int n = 64;
    unsigned long long *words; unsigned *turn, *xcc; long long *ticks; int *lost;
    hipMalloc(&words, sizeof(unsigned long long) * 32 * n);
    hipMalloc(&turn, 4); hipMalloc(&xcc, 4 * n); hipMalloc(&ticks, 8 * n); hipMalloc(&lost, 4);
    long long *rounds; hipMalloc(&rounds, 16 * n);
    // the same four modes with the exchanged words in fine-grained, then in uncached device memory (hipExtMallocWithFlags)
    unsigned long long *words_plain = words, *words_fg = nullptr, *words_uc = nullptr;
    if (hipExtMallocWithFlags((void **)&words_fg, sizeof(unsigned long long) * 32 * n, hipDeviceMallocFinegrained) != hipSuccess) words_fg = nullptr;
    if (hipExtMallocWithFlags((void **)&words_uc, sizeof(unsigned long long) * 32 * n, hipDeviceMallocUncached) != hipSuccess) words_uc = nullptr;
    (void)hipGetLastError();
    for (int alloc = 0; alloc < 3; ++alloc) {
    words = alloc == 0 ? words_plain : alloc == 1 ? words_fg : words_uc;
    if (!words) { printf("allocation kind %d not available\n", alloc); continue; }
    printf("--- words in %s memory\n", alloc == 0 ? "ordinary device" : alloc == 1 ? "fine-grained device" : "uncached device");
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(words, 0, sizeof(unsigned long long) * 32 * n);
        hipMemset(turn, 0, 4); hipMemset(ticks, 0, 8 * n); hipMemset(lost, 0, 4); hipMemset(rounds, 0, 16 * n);
        if (mode == 0) hipLaunchKernelGGL(pingpong<0>, dim3(n), dim3(64), 0, 0, words, turn, xcc, ticks, lost, rounds);
        if (mode == 1) hipLaunchKernelGGL(pingpong<1>, dim3(n), dim3(64), 0, 0, words, turn, xcc, ticks, lost, rounds);
        if (mode == 2) hipLaunchKernelGGL(pingpong<2>, dim3(n), dim3(64), 0, 0, words, turn, xcc, ticks, lost, rounds);
        if (mode == 3) hipLaunchKernelGGL(pingpong<3>, dim3(n), dim3(64), 0, 0, words, turn, xcc, ticks, lost, rounds);
        const hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned> hx(n); std::vector<long long> ht(n), hr(2 * n); int hl = 0;
        hipMemcpy(hx.data(), xcc, 4 * n, hipMemcpyDeviceToHost);
        hipMemcpy(ht.data(), ticks, 8 * n, hipMemcpyDeviceToHost);
        hipMemcpy(&hl, lost, 4, hipMemcpyDeviceToHost);
        hipMemcpy(hr.data(), rounds, 16 * n, hipMemcpyDeviceToHost);
        printf("mode %d (%s): ping on XCD %u%s%s\n", mode, hipGetErrorString(e), hx[0], hl ? ", LOST hand-off with partner " : "", hl ? std::to_string(hl).c_str() : "");
        double same = 0, other = 0; int ns = 0, no = 0;
        double rs = 0, cs = 0, ro = 0, co = 0;
        for (int p = 1; p < n; ++p) {
            if (!ht[p]) continue;
            if (hx[p] == hx[0]) { rs += (double)hr[2 * p]; cs += (double)hr[2 * p + 1]; } else { ro += (double)hr[2 * p]; co += (double)hr[2 * p + 1]; }
            const double us = (double)ht[p] / 100.0 / kTrips / 2.0; // one way, s_memrealtime = 100 MHz
            if (hx[p] == hx[0]) { same += us; ++ns; } else { other += us; ++no; }
        }
        printf("  one-way hand-off: same XCD %.3f us (%d partners), other XCD %.3f us (%d partners)\n", ns ? same / ns : 0.0, ns, no ? other / no : 0.0, no);
        printf("  consumer's poll round: same XCD %.0f core cycles (%.1f rounds per trip), other XCD %.0f (%.1f)\n", rs ? cs / rs : 0.0, ns ? rs / ns / kTrips : 0.0,
               ro ? co / ro : 0.0, no ? ro / no / kTrips : 0.0);
    }
    }
    return 0;
}
