// step_bench.hip -- what one step of the tiled sweep's compute wave costs when nothing else runs: a single wave
// walks a chain of dependent steps (codes and values from LDS, operands from LDS, fma chain, division, LDS store,
// two global stores), in the variants the kernel could take.  Core cycles per step.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sb tools/step_bench.hip && /tmp/sb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kSteps = 2000;

// correctly rounded n / d from a reciprocal refined as the compiler's own expansion of the division does it
// (v_rcp_f64, two Newton steps); valid while neither operand needs v_div_scale's rescaling
__device__ __forceinline__ double refined_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    return fma(r, e, r);
}
__device__ __forceinline__ double div_by(double n, double d, double r) {
    const double q = n * r;
    const double rem = fma(-d, q, n);
    return fma(rem, r, q);
}

// VARIANT bit 0: global stores, bit 1: division through the refined reciprocal (computed under the operand reads),
// bit 2: codes / values / row operands of step s+1 read during step s
template <int VARIANT, int NQ>
__global__ __launch_bounds__(256) void steps(double *x, unsigned long long *xs, long long *cycles, double *sink) {
    __shared__ unsigned long long opnd[1026];
    __shared__ int4 code[128];
    __shared__ double2 val[256];
    __shared__ int row[256];
    __shared__ double2 bD[256];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1026; i += 256) opnd[i] = (unsigned long long)__double_as_longlong(1.0 + 1e-3 * i);
    for (int i = threadIdx.x; i < 128; i += 256) code[i] = make_int4((i * 7) & 511, (i * 13 + 1) & 511, (i * 29 + 2) & 511, (i * 31 + 3) & 511);
    x += (size_t)blockIdx.x * 4096; xs += (size_t)blockIdx.x * 4096; sink += (size_t)blockIdx.x * 65;
    for (int i = threadIdx.x; i < 256; i += 256) { val[i] = make_double2(1e-3, -2e-3); row[i] = i * 97 % 4096; bD[i] = make_double2(0.5 + i, 3.0 + 1e-2 * i); }
    __shared__ unsigned done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (threadIdx.x >= 64) {
        if (!(VARIANT & 8)) return;
        // helper waves as in the sweep: two spin on an LDS word (s_sleep 2), one polls memory (two scattered loads, s_sleep 1)
        const int wave = threadIdx.x >> 6;
        unsigned long long seen = 0;
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
            if (wave == 3) {
                seen += __hip_atomic_load(&xs[(lane * 37) & 4095], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                seen += __hip_atomic_load(&xs[(lane * 53 + 7) & 4095], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_sleep(1);
            } else __builtin_amdgcn_s_sleep(2);
        }
        if (seen == 12345) sink[64] = 1.0;
        return;
    }
    double total = 0.0;
    const long long c0 = (long long)__builtin_readcyclecounter();
    int4 c_n[NQ]; double2 va_n[NQ], vb_n[NQ]; int row_n = 0; double2 bd_n = make_double2(0.0, 1.0);
    auto fetch = [&](int s, int4 *c, double2 *va, double2 *vb, int &r, double2 &bd) {
        const int sl = (s * 64 + lane) & 255;
        r = row[sl]; bd = bD[sl];
#pragma unroll
        for (int u = 0; u < NQ; ++u) { const int q = (s * 64 * NQ + u * 64 + lane) & 127; c[u] = code[q]; va[u] = val[2 * q]; vb[u] = val[2 * q + 1]; }
    };
    if (VARIANT & 4) fetch(0, c_n, va_n, vb_n, row_n, bd_n);
    for (int s = 0; s < kSteps; ++s) {
        int4 c[NQ]; double2 va[NQ], vb[NQ]; int r; double2 bd;
        if (VARIANT & 4) {
#pragma unroll
            for (int u = 0; u < NQ; ++u) { c[u] = c_n[u]; va[u] = va_n[u]; vb[u] = vb_n[u]; }
            r = row_n; bd = bd_n;
        } else fetch(s, c, va, vb, r, bd);
        unsigned long long xo[NQ][4];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            xo[u][0] = __hip_atomic_load(&opnd[c[u].x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            xo[u][1] = __hip_atomic_load(&opnd[c[u].y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            xo[u][2] = __hip_atomic_load(&opnd[c[u].z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            xo[u][3] = __hip_atomic_load(&opnd[c[u].w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (VARIANT & 4) fetch(s + 1, c_n, va_n, vb_n, row_n, bd_n);
        double rcp = 0.0;
        if (VARIANT & 2) rcp = refined_rcp(bd.y);
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            acc = fma(va[u].x, __longlong_as_double((long long)xo[u][0]), acc);
            acc = fma(va[u].y, __longlong_as_double((long long)xo[u][1]), acc);
            acc = fma(vb[u].x, __longlong_as_double((long long)xo[u][2]), acc);
            acc = fma(vb[u].y, __longlong_as_double((long long)xo[u][3]), acc);
        }
        const double n = bd.x - acc;
        const double res = (VARIANT & 2) ? div_by(n, bd.y, rcp) : n / bd.y;
        const unsigned long long out = (unsigned long long)__double_as_longlong(res);
        __hip_atomic_store(&opnd[(s * 64 + lane) & 511], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (VARIANT & 1) {
            x[r] = res;
            __hip_atomic_store(&xs[(s * 64 + lane) & 4095], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        total += res;
    }
    if (lane == 0) { cycles[blockIdx.x] = (long long)__builtin_readcyclecounter() - c0; __hip_atomic_store(&done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    sink[lane] = total;
}

constexpr int kMaxGrid = 2048;
template <int V, int NQ>
static void run(const char *what, double *x, unsigned long long *xs, long long *cyc, double *sink, int grid = 1) {
    hipLaunchKernelGGL((steps<V, NQ>), dim3(grid), dim3(256), 0, 0, x, xs, cyc, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((steps<V, NQ>), dim3(grid), dim3(256), 0, 0, x, xs, cyc, sink);
    hipDeviceSynchronize();
    std::vector<long long> h(grid); double hs[64];
    hipMemcpy(h.data(), cyc, 8 * grid, hipMemcpyDeviceToHost);
    hipMemcpy(hs, sink, sizeof(hs), hipMemcpyDeviceToHost);
    double sum = 0; for (long long v : h) sum += (double)v;
    printf("%d quad(s) per row, %4d workgroups, %-52s %6.0f cycles per step   (checksum %.17g)\n", NQ, grid, what, sum / grid / kSteps, hs[5]);
}

int main() {
    double *x, *sink; unsigned long long *xs; long long *cyc;
    hipMalloc(&x, 8 * 4096 * kMaxGrid); hipMalloc(&xs, 8 * 4096 * kMaxGrid); hipMalloc(&cyc, 8 * kMaxGrid); hipMalloc(&sink, 8 * 65 * kMaxGrid);
    run<0, 1>("as today, no global stores:", x, xs, cyc, sink);
    run<1, 1>("as today:", x, xs, cyc, sink);
    run<3, 1>("reciprocal division:", x, xs, cyc, sink);
    run<5, 1>("next step's codes prefetched:", x, xs, cyc, sink);
    run<7, 1>("prefetched + reciprocal division:", x, xs, cyc, sink);
    run<9, 1>("as today, helper waves spinning beside it:", x, xs, cyc, sink);
    run<15, 1>("prefetched + reciprocal, helpers spinning:", x, xs, cyc, sink);
    run<9, 4>("as today, helper waves spinning beside it:", x, xs, cyc, sink);
    run<1, 4>("as today:", x, xs, cyc, sink);
    run<3, 4>("reciprocal division:", x, xs, cyc, sink);
    run<5, 4>("next step's codes prefetched:", x, xs, cyc, sink);
    run<7, 4>("prefetched + reciprocal division:", x, xs, cyc, sink);
    for (int per_cu : {1, 2, 4, 8}) {
        run<9, 1>("as today + helpers, chip full:", x, xs, cyc, sink, 256 * per_cu);
        run<8, 1>("no global stores + helpers, chip full:", x, xs, cyc, sink, 256 * per_cu);
    }
    return 0;
}
