// tools/blas1_bench.hip -- streaming-kernel variants for the axpy-class / dot kernels (round 5).
//   hipcc -O3 --offload-arch=gfx950 tools/blas1_bench.hip -o tools/blas1_bench && tools/blas1_bench [N] [reps]
// Triad r = a + s b (24 N bytes), copy (16 N), dot (16 N), sumsq (8 N): GB/s per variant, HIP-event timed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

typedef double v2d __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// A: the library's current form: double2, grid-stride, plain loads / stores
__global__ __launch_bounds__(256) void triad_a(double *r, const double *a, const double *b, int64_t n, double s) {
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * 256;
    const v2d *a2 = (const v2d *)a, *b2 = (const v2d *)b; v2d *r2 = (v2d *)r;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) { v2d av = a2[i], bv = b2[i]; r2[i] = av + s * bv; }
}
// B: non-temporal loads and stores
__global__ __launch_bounds__(256) void triad_b(double *r, const double *a, const double *b, int64_t n, double s) {
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * 256;
    const v2d *a2 = (const v2d *)a, *b2 = (const v2d *)b; v2d *r2 = (v2d *)r;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
        v2d av = __builtin_nontemporal_load(a2 + i), bv = __builtin_nontemporal_load(b2 + i);
        __builtin_nontemporal_store(av + s * bv, r2 + i);
    }
}
// C<U, NT>: U independent 16-byte accesses per operand in flight per lane (grid-stride in units of U * grid), optional nt
template <int U, int NT>
__global__ __launch_bounds__(256) void triad_c(double *r, const double *a, const double *b, int64_t n, double s) {
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * 256;
    const v2d *a2 = (const v2d *)a, *b2 = (const v2d *)b; v2d *r2 = (v2d *)r;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        v2d av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            av[u] = (NT & 1) ? __builtin_nontemporal_load(a2 + i + u * stride) : a2[i + u * stride];
            bv[u] = (NT & 1) ? __builtin_nontemporal_load(b2 + i + u * stride) : b2[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT & 2) __builtin_nontemporal_store(av[u] + s * bv[u], r2 + i + u * stride); else r2[i + u * stride] = av[u] + s * bv[u];
        }
    }
    for (; i < n2; i += stride) r2[i] = a2[i] + s * b2[i];
}
// D<U, NT>: every workgroup owns ONE contiguous range (block-contiguous instead of grid-stride), U accesses in flight
template <int U, int NT>
__global__ __launch_bounds__(256) void triad_d(double *r, const double *a, const double *b, int64_t n, double s) {
    const int64_t n2 = n >> 1;
    const int64_t per = ((n2 + gridDim.x - 1) / gridDim.x + 255) & ~(int64_t)255;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    const v2d *a2 = (const v2d *)a, *b2 = (const v2d *)b; v2d *r2 = (v2d *)r;
    int64_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        v2d av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            av[u] = (NT & 1) ? __builtin_nontemporal_load(a2 + i + u * 256) : a2[i + u * 256];
            bv[u] = (NT & 1) ? __builtin_nontemporal_load(b2 + i + u * 256) : b2[i + u * 256];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT & 2) __builtin_nontemporal_store(av[u] + s * bv[u], r2 + i + u * 256); else r2[i + u * 256] = av[u] + s * bv[u];
        }
    }
    for (; i < hi; i += 256) r2[i] = a2[i] + s * b2[i];
}
// sum of squares (read-only): U loads in flight, optional nt; partial per block
template <int U, int NT>
__global__ __launch_bounds__(256) void sumsq_c(const double *a, int64_t n, double *partials) {
    __shared__ double lds[4];
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * 256;
    const v2d *a2 = (const v2d *)a;
    double acc0 = 0.0, acc1 = 0.0;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        v2d av[U];
#pragma unroll
        for (int u = 0; u < U; ++u) av[u] = NT ? __builtin_nontemporal_load(a2 + i + u * stride) : a2[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc0 = fma(av[u].x, av[u].x, acc0); acc1 = fma(av[u].y, av[u].y, acc1); }
    }
    for (; i < n2; i += stride) { v2d av = a2[i]; acc0 = fma(av.x, av.x, acc0); acc1 = fma(av.y, av.y, acc1); }
    double v = acc0 + acc1;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}
template <int U, int NT>
__global__ __launch_bounds__(256) void dot_c(const double *a, const double *b, int64_t n, double *partials) {
    __shared__ double lds[4];
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * 256;
    const v2d *a2 = (const v2d *)a, *b2 = (const v2d *)b;
    double acc0 = 0.0, acc1 = 0.0;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        v2d av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { av[u] = NT ? __builtin_nontemporal_load(a2 + i + u * stride) : a2[i + u * stride];
                                      bv[u] = NT ? __builtin_nontemporal_load(b2 + i + u * stride) : b2[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc0 = fma(av[u].x, bv[u].x, acc0); acc1 = fma(av[u].y, bv[u].y, acc1); }
    }
    for (; i < n2; i += stride) { v2d av = a2[i], bv = b2[i]; acc0 = fma(av.x, bv.x, acc0); acc1 = fma(av.y, bv.y, acc1); }
    double v = acc0 + acc1;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}
template <int U, int NT>
__global__ __launch_bounds__(256) void copy_c(double *r, const double *a, int64_t n) {
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * 256;
    const v2d *a2 = (const v2d *)a; v2d *r2 = (v2d *)r;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        v2d av[U];
#pragma unroll
        for (int u = 0; u < U; ++u) av[u] = (NT & 1) ? __builtin_nontemporal_load(a2 + i + u * stride) : a2[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT & 2) __builtin_nontemporal_store(av[u], r2 + i + u * stride); else r2[i + u * stride] = av[u]; }
    }
    for (; i < n2; i += stride) r2[i] = a2[i];
}

int main(int argc, char **argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 16777216;
    const int reps = argc > 2 ? atoi(argv[2]) : 50;
    double *a, *b, *r, *part, *spare[6];
    CHECK(hipMalloc(&a, 8 * N)); CHECK(hipMalloc(&b, 8 * N)); CHECK(hipMalloc(&r, 8 * N)); CHECK(hipMalloc(&part, 8 * 65536));
    for (int k = 0; k < 6; ++k) { CHECK(hipMalloc(&spare[k], 8 * N)); CHECK(hipMemset(spare[k], 0, 8 * N)); }
    std::vector<double> h(N, 1.0);
    CHECK(hipMemcpy(a, h.data(), 8 * N, hipMemcpyHostToDevice)); CHECK(hipMemcpy(b, h.data(), 8 * N, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, double bytes, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        CHECK(hipDeviceSynchronize());
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.1f us  %7.1f GB/s\n", name, 1e3 * ms / reps, bytes * reps / ms / 1e6);
    };
    const double T = 24.0 * N, Cp = 16.0 * N, Rd = 8.0 * N;
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        printf("-- grid %d\n", grid);
        char nm[64];
#define RUN(label, bytes, ...) snprintf(nm, sizeof nm, "%s g%d", label, grid); timeit(nm, bytes, [&] { __VA_ARGS__; })
        RUN("triad A plain", T, hipLaunchKernelGGL(triad_a, dim3(grid), dim3(256), 0, 0, r, a, b, N, 0.5));
        RUN("triad B nt", T, hipLaunchKernelGGL(triad_b, dim3(grid), dim3(256), 0, 0, r, a, b, N, 0.5));
        RUN("triad C u2", T, hipLaunchKernelGGL((triad_c<2, 0>), dim3(grid), dim3(256), 0, 0, r, a, b, N, 0.5));
        RUN("triad C u4", T, hipLaunchKernelGGL((triad_c<4, 0>), dim3(grid), dim3(256), 0, 0, r, a, b, N, 0.5));
        RUN("triad C u2 nt-ld", T, hipLaunchKernelGGL((triad_c<2, 1>), dim3(grid), dim3(256), 0, 0, r, a, b, N, 0.5));
        RUN("triad C u2 nt-st", T, hipLaunchKernelGGL((triad_c<2, 2>), dim3(grid), dim3(256), 0, 0, r, a, b, N, 0.5));
        RUN("triad C u2 nt", T, hipLaunchKernelGGL((triad_c<2, 3>), dim3(grid), dim3(256), 0, 0, r, a, b, N, 0.5));
        RUN("triad C u4 nt", T, hipLaunchKernelGGL((triad_c<4, 3>), dim3(grid), dim3(256), 0, 0, r, a, b, N, 0.5));
        RUN("triad D contiguous u2", T, hipLaunchKernelGGL((triad_d<2, 0>), dim3(grid), dim3(256), 0, 0, r, a, b, N, 0.5));
        RUN("triad D contiguous u4 nt", T, hipLaunchKernelGGL((triad_d<4, 3>), dim3(grid), dim3(256), 0, 0, r, a, b, N, 0.5));
        RUN("copy u1", Cp, hipLaunchKernelGGL((copy_c<1, 0>), dim3(grid), dim3(256), 0, 0, r, a, N));
        RUN("copy u4", Cp, hipLaunchKernelGGL((copy_c<4, 0>), dim3(grid), dim3(256), 0, 0, r, a, N));
        RUN("copy u4 nt", Cp, hipLaunchKernelGGL((copy_c<4, 3>), dim3(grid), dim3(256), 0, 0, r, a, N));
        if (grid <= 4096) {
            RUN("sumsq u1", Rd, hipLaunchKernelGGL((sumsq_c<1, 0>), dim3(grid), dim3(256), 0, 0, a, N, part));
            RUN("sumsq u4", Rd, hipLaunchKernelGGL((sumsq_c<4, 0>), dim3(grid), dim3(256), 0, 0, a, N, part));
            RUN("sumsq u4 nt", Rd, hipLaunchKernelGGL((sumsq_c<4, 1>), dim3(grid), dim3(256), 0, 0, a, N, part));
            RUN("sumsq u8 nt", Rd, hipLaunchKernelGGL((sumsq_c<8, 1>), dim3(grid), dim3(256), 0, 0, a, N, part));
        }
    }
    // ROTATING operands: 8 vectors, every launch reads and writes other ones than the launch before (no reuse through the
    // 256 MB Infinity Cache: the honest streaming figure)
    {
        double *v[8]; for (int k = 0; k < 6; ++k) v[k] = spare[k]; v[6] = a; v[7] = b;
        for (int grid : {2048, 4096, 8192, 16384}) {
            printf("-- rotating operands, grid %d\n", grid);
            char nm[64]; int k;
#define ROT(label, bytes, ...) k = 0; snprintf(nm, sizeof nm, "rot %s g%d", label, grid); timeit(nm, bytes, [&] { double *R = v[k % 8], *A = v[(k + 3) % 8], *B = v[(k + 6) % 8]; (void)B; __VA_ARGS__; k += 1; })
            ROT("triad plain", T, hipLaunchKernelGGL(triad_a, dim3(grid), dim3(256), 0, 0, R, A, B, N, 0.5));
            ROT("triad nt both u1", T, hipLaunchKernelGGL(triad_b, dim3(grid), dim3(256), 0, 0, R, A, B, N, 0.5));
            ROT("triad nt-st u1", T, hipLaunchKernelGGL((triad_c<1, 2>), dim3(grid), dim3(256), 0, 0, R, A, B, N, 0.5));
            ROT("triad nt-st u2", T, hipLaunchKernelGGL((triad_c<2, 2>), dim3(grid), dim3(256), 0, 0, R, A, B, N, 0.5));
            ROT("triad nt-ld u1", T, hipLaunchKernelGGL((triad_c<1, 1>), dim3(grid), dim3(256), 0, 0, R, A, B, N, 0.5));
            ROT("triad contiguous u2 nt-st", T, hipLaunchKernelGGL((triad_d<2, 2>), dim3(grid), dim3(256), 0, 0, R, A, B, N, 0.5));
            ROT("copy plain", Cp, hipLaunchKernelGGL((copy_c<1, 0>), dim3(grid), dim3(256), 0, 0, R, A, N));
            ROT("copy nt-st", Cp, hipLaunchKernelGGL((copy_c<1, 2>), dim3(grid), dim3(256), 0, 0, R, A, N));
            ROT("copy nt both", Cp, hipLaunchKernelGGL((copy_c<1, 3>), dim3(grid), dim3(256), 0, 0, R, A, N));
            ROT("dot plain", Cp, hipLaunchKernelGGL((dot_c<1, 0>), dim3(grid), dim3(256), 0, 0, A, B, N, part));
            ROT("dot nt", Cp, hipLaunchKernelGGL((dot_c<1, 1>), dim3(grid), dim3(256), 0, 0, A, B, N, part));
            ROT("dot nt u2", Cp, hipLaunchKernelGGL((dot_c<2, 1>), dim3(grid), dim3(256), 0, 0, A, B, N, part));
            if (grid <= 4096) {
                ROT("sumsq u1", Rd, hipLaunchKernelGGL((sumsq_c<1, 0>), dim3(grid), dim3(256), 0, 0, A, N, part));
                ROT("sumsq u1 nt", Rd, hipLaunchKernelGGL((sumsq_c<1, 1>), dim3(grid), dim3(256), 0, 0, A, N, part));
                ROT("sumsq u2", Rd, hipLaunchKernelGGL((sumsq_c<2, 0>), dim3(grid), dim3(256), 0, 0, A, N, part));
            }
        }
    }
    return 0;
}
