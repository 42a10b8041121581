// bis_reorder.hip -- multi-colour symmetric reordering on the device (the role of
// SMAX's PERM_MODE in the reference: CMakeLists.txt:128-133,
// utilities/smax_helpers.hpp:44-80 `permute_mat`).  B = P A P^T with the rows
// grouped by colour, so that the strict triangles of B have as many dependency
// levels as there are colours and the Gauss-Seidel / ILU sweeps run at SpMV
// speed (bis_sptrsv.hip, few-level path).
//
//   colouring  : greedy first-fit in NATURAL row order -- the sequential
//                algorithm, row r takes the smallest colour not used by its
//                lower-numbered neighbours -- executed in ONE launch: a
//                persistent grid takes 256-row tickets in row order and every
//                lane polls the colours of its lower neighbours (a sentinel
//                marks "not yet coloured"; the published colour is the flag).
//                Ticket order guarantees progress; lanes of one wave depend on
//                each other (r on r-1), so the publishing store is predicated
//                inside volatile asm on the straight-line path of the wait
//                loop (see the hazard note in bis_sptrsv.hip).
//   permutation: stable counting sort of the rows by colour (perm[new] = old).
//   B          : row lengths gathered through perm, scanned, rows copied with
//                their entries in the original order and columns renumbered
//                through the inverse permutation (same as the host version in
//                host/utilities/permute.hpp, which remains as the fallback).
//
// The colouring looks at the entries of row r with column < r, i.e. it assumes
// a structurally symmetric pattern; on an unsymmetric pattern two coupled rows
// may share a colour, which costs levels, never correctness (the triangular
// solves derive their levels from the permuted matrix itself).
#include "bis_internal.hpp"

#include <algorithm>

namespace {

constexpr int kMaxColours = 64;
constexpr unsigned kColSpinLimit = 1u << 22;
constexpr int kColBatch = 8;

template <typename RP>
__global__ __launch_bounds__(256) void greedy_colour_kernel(const RP *__restrict__ row_ptr,
                                                            const int32_t *__restrict__ col, int64_t n,
                                                            int *colour /* -1 = not yet */, unsigned *ticket,
                                                            int *status /* [0] = needs > 64 colours or lost hand-off */) {
    __shared__ unsigned s_ticket;
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        const unsigned t = s_ticket;
        __syncthreads();
        const int64_t base = (int64_t)t * 256;
        if (base >= n) return;
        const bool valid = base + threadIdx.x < n; // lanes past the end run an empty row (no divergent branch at the loop tail)
        const int64_t r = valid ? base + threadIdx.x : 0;
        {
            int64_t k = valid ? (int64_t)row_ptr[r] : 0;
            const int64_t e = valid ? (int64_t)row_ptr[r + 1] : 0;
            unsigned long long used = 0ull;
            bool overflow = false, done = false;
            unsigned spins = 0;
            int cv[kColBatch], pc[kColBatch];
#pragma unroll
            for (int j = 0; j < kColBatch; ++j) { cv[j] = 0; pc[j] = 0; }
            int ready = 0, in_batch = 0;
            while (!done) { // one loop, bounded work per trip (lanes of a wave wait for each other)
                bool publish = false;
                int out = 0;
                if (ready == in_batch) {
                    k += in_batch;
                    if (k == e) {
                        const unsigned long long free_mask = ~used;
                        out = free_mask ? __builtin_ctzll(free_mask) : kMaxColours - 1;
                        if (!free_mask) overflow = true;
                        publish = true;
                        in_batch = ready = 0;
                    } else {
                        in_batch = e - k < (int64_t)kColBatch ? (int)(e - k) : kColBatch;
#pragma unroll
                        for (int j = 0; j < kColBatch; ++j) pc[j] = j < in_batch ? col[k + j] : 0;
#pragma unroll
                        for (int j = 0; j < kColBatch; ++j) // entries at or above the diagonal do not constrain r
                            cv[j] = (j < in_batch && pc[j] < r) ? __hip_atomic_load(&colour[pc[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                                : 1 << 30;
                        ready = 0;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < kColBatch; ++j)
                        if (j >= ready && j < in_batch && cv[j] < 0)
                            cv[j] = __hip_atomic_load(&colour[pc[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < kColBatch; ++j) {
                    if (ready == j && j < in_batch && cv[j] >= 0) {
                        if (cv[j] < kMaxColours) used |= 1ull << cv[j];
                        ready = j + 1;
                    }
                }
                if (ready < in_batch) {
                    if (++spins > kColSpinLimit) { publish = true; overflow = true; out = 0; }
                    else __builtin_amdgcn_s_sleep(1);
                }
                {
                    int *dst = &colour[r];
                    const unsigned pflag = (publish && valid) ? 1u : 0u;
                    unsigned long long saved_exec;
                    asm volatile("v_cmp_ne_u32_e32 vcc, 0, %3\n\ts_and_saveexec_b64 %0, vcc\n\t"
                                 "global_store_dword %1, %2, off sc1\n\ts_mov_b64 exec, %0"
                                 : "=&s"(saved_exec) : "v"(dst), "v"(out), "v"(pflag) : "vcc", "memory");
                }
                if (publish) done = true;
            }
            // wave-uniform condition: no divergent branch at the tail of the ticket loop
            if (__any(overflow)) atomicOr(status, overflow ? 1 : 0);
        }
    }
}

// per-block histogram of the colours
__global__ __launch_bounds__(256) void colour_hist_kernel(const int *__restrict__ colour, int64_t n, int n_col,
                                                          int64_t *__restrict__ hist /* [n_blk][n_col] */) {
    __shared__ int h[kMaxColours];
    if (threadIdx.x < kMaxColours) h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < n) atomicAdd(&h[colour[r]], 1);
    __syncthreads();
    if ((int)threadIdx.x < n_col) hist[(int64_t)blockIdx.x * n_col + threadIdx.x] = h[threadIdx.x];
}

// exclusive scan down the blocks for every colour (thread c owns colour c), then the
// colour bases: start of colour c = rows of all smaller colours
__global__ __launch_bounds__(kMaxColours) void colour_scan_kernel(int64_t *hist, int n_blk, int n_col,
                                                                  int64_t *colour_start /* [n_col+1] */) {
    __shared__ int64_t tot[kMaxColours];
    const int c = threadIdx.x;
    int64_t run = 0;
    if (c < n_col)
        for (int b = 0; b < n_blk; ++b) {
            const int64_t v = hist[(int64_t)b * n_col + c];
            hist[(int64_t)b * n_col + c] = run;
            run += v;
        }
    tot[c] = c < n_col ? run : 0;
    __syncthreads();
    if (c == 0) {
        int64_t s = 0;
        for (int i = 0; i < n_col; ++i) { colour_start[i] = s; s += tot[i]; }
        colour_start[n_col] = s;
    }
}

// stable scatter: position of row r = start of its colour + rows of that colour in
// earlier blocks + rows of that colour earlier in this block
__global__ __launch_bounds__(256) void colour_scatter_kernel(const int *__restrict__ colour, int64_t n, int n_col,
                                                             const int64_t *__restrict__ hist,
                                                             const int64_t *__restrict__ colour_start,
                                                             int32_t *__restrict__ perm, int32_t *__restrict__ inv) {
    __shared__ int sc[256];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int mine = r < n ? colour[r] : -1;
    for (int c = 0; c < n_col; ++c) {
        const int f = mine == c ? 1 : 0;
        sc[threadIdx.x] = f;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int a = 0;
            if ((int)threadIdx.x >= off) a = sc[threadIdx.x - off];
            __syncthreads();
            sc[threadIdx.x] += a;
            __syncthreads();
        }
        if (f) {
            const int64_t p = colour_start[c] + hist[(int64_t)blockIdx.x * n_col + c] + sc[threadIdx.x] - 1;
            perm[p] = (int32_t)r;
            inv[r] = (int32_t)p;
        }
        __syncthreads();
    }
}

// row lengths of B (new row i = old row perm[i]); per-block sums for the scan
template <typename RP>
__global__ __launch_bounds__(256) void perm_count_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ perm,
                                                         int64_t n, int64_t *__restrict__ blk) {
    __shared__ double lds[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int len = 0;
    if (i < n) { const int o = perm[i]; len = (int)(row_ptr[o + 1] - row_ptr[o]); }
    const double tot = block_sum<256>((double)len, lds);
    if (threadIdx.x == 0) blk[blockIdx.x] = (int64_t)tot;
}

// exclusive scan of the block sums (single workgroup)
__global__ __launch_bounds__(256) void blk_scan_kernel(int64_t *blk, int n_blk) {
    __shared__ int64_t s[256];
    int64_t run = 0;
    for (int base = 0; base < n_blk; base += 256) {
        const int i = base + threadIdx.x;
        const int64_t v = i < n_blk ? blk[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int64_t a = 0;
            if ((int)threadIdx.x >= off) a = s[threadIdx.x - off];
            __syncthreads();
            s[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < n_blk) blk[i] = run + s[threadIdx.x] - v;
        run += s[255];
        __syncthreads();
    }
}

template <typename RP>
__global__ __launch_bounds__(256) void perm_fill_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                        const double *__restrict__ val,
                                                        const int32_t *__restrict__ perm,
                                                        const int32_t *__restrict__ inv, int64_t n,
                                                        const int64_t *__restrict__ blk, RP *__restrict__ rpB,
                                                        int32_t *__restrict__ colB, double *__restrict__ valB) {
    __shared__ int64_t sc[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t a = 0;
    int len = 0;
    if (i < n) { const int o = perm[i]; a = (int64_t)row_ptr[o]; len = (int)((int64_t)row_ptr[o + 1] - a); }
    sc[threadIdx.x] = len;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int64_t v = 0;
        if ((int)threadIdx.x >= off) v = sc[threadIdx.x - off];
        __syncthreads();
        sc[threadIdx.x] += v;
        __syncthreads();
    }
    if (i >= n) return;
    const int64_t p = blk[blockIdx.x] + sc[threadIdx.x] - len;
    rpB[i] = (RP)p;
    if (i == n - 1) rpB[n] = (RP)(p + len);
    for (int q = 0; q < len; ++q) { // entries keep their order inside the row
        colB[p + q] = inv[col[a + q]];
        valB[p + q] = val[a + q];
    }
}

__global__ __launch_bounds__(256) void fill_int_kernel(int *p, int64_t n, int v) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = v;
}

__global__ __launch_bounds__(256) void max_int_kernel(const int *p, int64_t n, int *out) {
    int m = -1;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = max(m, p[i]);
    atomicMax(out, m);
}

__global__ __launch_bounds__(256) void gather_vec_kernel(const double *__restrict__ in, const int32_t *__restrict__ perm,
                                                         int64_t n, double *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = in[perm[i]];
}

__global__ __launch_bounds__(256) void scatter_vec_kernel(const double *__restrict__ in, const int32_t *__restrict__ perm,
                                                          int64_t n, double *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[perm[i]] = in[i];
}

// -scale on the device: extract_scale (utilities/LU_factors.hpp:880-898) + scale_mat
// (preprocessing.hpp:15-24).  s_r = 1/sqrt(|a_rr|) (the last diagonal entry of the row wins,
// rows without one keep the caller's value), then a_rc *= (s_r * s_c).
template <typename RP>
__global__ __launch_bounds__(256) void extract_scale_kernel(const RP *__restrict__ row_ptr,
                                                            const int32_t *__restrict__ col,
                                                            const double *__restrict__ val, int64_t n,
                                                            double *__restrict__ s, unsigned long long *status) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    for (int64_t k = (int64_t)row_ptr[r]; k < (int64_t)row_ptr[r + 1]; ++k)
        if (col[k] == r) {
            const double v = val[k];
            if (fabs(v) < 1e-16) atomicMin(status, (unsigned long long)(r + 1));
            s[r] = 1.0 / sqrt(fabs(v));
        }
}
template <typename RP>
__global__ __launch_bounds__(256) void scale_mat_kernel(const RP *__restrict__ row_ptr,
                                                        const int32_t *__restrict__ col, double *__restrict__ val,
                                                        int64_t n, const double *__restrict__ s) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const double sr = s[r];
    for (int64_t k = (int64_t)row_ptr[r]; k < (int64_t)row_ptr[r + 1]; ++k) val[k] *= (sr * s[col[k]]);
}

template <typename RP>
bis_status multicolour_t(bis_ctx *ctx, const bis_mat *A, bis_mat **B_out, int32_t *perm_dev, int *n_colours_out) {
    const int64_t n = A->n_rows;
    const RP *rp = (const RP *)A->row_ptr;
    const int n_blk = (int)((n + 255) / 256);
    int *colour = nullptr, *status = nullptr;
    unsigned *ticket = nullptr;
    int64_t *hist = nullptr, *blk = nullptr, *cstart = nullptr;
    int32_t *inv = nullptr;
    bis_mat *B = nullptr;
    auto cleanup = [&](bis_status rc) {
        hipFree(colour); hipFree(status); hipFree(ticket); hipFree(hist); hipFree(blk); hipFree(cstart); hipFree(inv);
        if (rc != BIS_OK && B) bis_mat_destroy(ctx, B);
        return rc;
    };
#define BIS_RO_CHECK(call)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return cleanup(BIS_ERR_HIP); } \
    } while (0)
    BIS_RO_CHECK(hipMalloc(&colour, sizeof(int) * (size_t)std::max<int64_t>(n, 1)));
    BIS_RO_CHECK(hipMalloc(&status, sizeof(int) * 4));
    BIS_RO_CHECK(hipMalloc(&ticket, sizeof(unsigned) * 4));
    BIS_RO_CHECK(hipMalloc(&inv, sizeof(int32_t) * (size_t)std::max<int64_t>(n, 1)));
    BIS_RO_CHECK(hipMalloc(&blk, sizeof(int64_t) * (size_t)(n_blk + 1)));
    BIS_RO_CHECK(hipMalloc(&cstart, sizeof(int64_t) * (kMaxColours + 1)));
    BIS_RO_CHECK(hipMemsetAsync(status, 0, sizeof(int) * 4, ctx->stream));
    BIS_RO_CHECK(hipMemsetAsync(ticket, 0, sizeof(unsigned) * 4, ctx->stream));
    hipLaunchKernelGGL(fill_int_kernel, dim3((unsigned)std::min<int64_t>(n_blk, 2048)), dim3(256), 0, ctx->stream, colour, n, -1);
    // persistent grid, resident by construction (a few workgroups per CU)
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_blk, (int64_t)ctx->n_cus * 4));
    hipLaunchKernelGGL(greedy_colour_kernel<RP>, dim3(grid), dim3(256), 0, ctx->stream, rp, A->col, n, colour, ticket, status);
    hipLaunchKernelGGL(max_int_kernel, dim3((unsigned)std::min<int64_t>(n_blk, 1024)), dim3(256), 0, ctx->stream, colour, n, status + 1);
    int h[2] = {0, 0};
    BIS_RO_CHECK(hipMemcpyAsync(h, status, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    BIS_RO_CHECK(hipStreamSynchronize(ctx->stream));
    if (h[0]) { ctx->err = "bis_mat_multicolour: more than 64 colours needed"; return cleanup(BIS_ERR_UNSUPPORTED); }
    const int n_col = h[1] + 1;
    BIS_RO_CHECK(hipMalloc(&hist, sizeof(int64_t) * (size_t)n_blk * n_col));
    hipLaunchKernelGGL(colour_hist_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, colour, n, n_col, hist);
    hipLaunchKernelGGL(colour_scan_kernel, dim3(1), dim3(kMaxColours), 0, ctx->stream, hist, n_blk, n_col, cstart);
    hipLaunchKernelGGL(colour_scatter_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, colour, n, n_col, hist, cstart,
                       perm_dev, inv);
    bis_status st = bis_mat_alloc(ctx, n, A->n_cols, A->nnz, A->rp64, &B);
    if (st != BIS_OK) return cleanup(st);
    hipLaunchKernelGGL(perm_count_kernel<RP>, dim3(n_blk), dim3(256), 0, ctx->stream, rp, perm_dev, n, blk);
    hipLaunchKernelGGL(blk_scan_kernel, dim3(1), dim3(256), 0, ctx->stream, blk, n_blk);
    hipLaunchKernelGGL(perm_fill_kernel<RP>, dim3(n_blk), dim3(256), 0, ctx->stream, rp, A->col, A->val, perm_dev, inv, n,
                       blk, (RP *)B->row_ptr, B->col, B->val);
    BIS_RO_CHECK(hipGetLastError());
    BIS_RO_CHECK(hipStreamSynchronize(ctx->stream));
#undef BIS_RO_CHECK
    st = bis_mat_finalize(ctx, B);
    if (st != BIS_OK) return cleanup(st);
    *B_out = B;
    if (n_colours_out) *n_colours_out = n_col;
    return cleanup(BIS_OK);
}

} // namespace

extern "C" {

bis_status bis_mat_multicolour(bis_ctx *ctx, const bis_mat *A, bis_mat **B, int32_t *perm_dev, int *n_colours) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A && B && perm_dev, "bis_mat_multicolour: bad arguments");
    BIS_REQUIRE(ctx, A->n_rows == A->n_cols, "bis_mat_multicolour: square matrix required");
    BIS_REQUIRE(ctx, !A->view, "bis_mat_multicolour: not for row views");
    if (A->n_rows == 0) { ctx->err = "bis_mat_multicolour: empty matrix"; return BIS_ERR_INVALID; }
    return A->rp64 ? multicolour_t<int64_t>(ctx, A, B, perm_dev, n_colours)
                   : multicolour_t<int32_t>(ctx, A, B, perm_dev, n_colours);
}


bis_status bis_mat_scale_sym(bis_ctx *ctx, bis_mat *A, double *scale) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A && scale && !A->view && A->n_rows == A->n_cols, "bis_mat_scale_sym: bad arguments");
    const int64_t n = A->n_rows;
    if (n == 0) return BIS_OK;
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); // sweeps in flight still read the plans dropped next
    bis_mat_values_changed(A); // the values change in place
    unsigned long long *status = (unsigned long long *)(ctx->scalars_dev + 32);
    BIS_HIP_CHECK(ctx, hipMemsetAsync(status, 0xFF, 8, ctx->stream));
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (A->rp64) {
        hipLaunchKernelGGL(extract_scale_kernel<int64_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int64_t *)A->row_ptr, A->col, A->val, n, scale, status);
        hipLaunchKernelGGL(scale_mat_kernel<int64_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int64_t *)A->row_ptr, A->col, A->val, n, scale);
    } else {
        hipLaunchKernelGGL(extract_scale_kernel<int32_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col, A->val, n, scale, status);
        hipLaunchKernelGGL(scale_mat_kernel<int32_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col, A->val, n, scale);
    }
    BIS_HIP_CHECK(ctx, hipGetLastError());
    unsigned long long h = 0;
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&h, status, 8, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h != ~0ull) {
        char msg[96];
        snprintf(msg, sizeof msg, "Zero detected on diagonal at row index %lld", (long long)h - 1);
        ctx->err = msg;
        return BIS_ERR_ZERO_DIAG;
    }
    return BIS_OK;
}


bis_status bis_vec_gather(bis_ctx *ctx, double *out, const double *in, const int32_t *perm_dev, int64_t n) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, n >= 0 && (n == 0 || (out && in && perm_dev)) && out != in, "bis_vec_gather: bad arguments");
    if (n == 0) return BIS_OK;
    hipLaunchKernelGGL(gather_vec_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0,
                       ctx->stream, in, perm_dev, n, out);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}

bis_status bis_vec_scatter(bis_ctx *ctx, double *out, const double *in, const int32_t *perm_dev, int64_t n) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, n >= 0 && (n == 0 || (out && in && perm_dev)) && out != in, "bis_vec_scatter: bad arguments");
    if (n == 0) return BIS_OK;
    hipLaunchKernelGGL(scatter_vec_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0,
                       ctx->stream, in, perm_dev, n, out);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}

} // extern "C"
