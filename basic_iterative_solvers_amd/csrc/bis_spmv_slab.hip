// bis_spmv_slab.hip -- column slabs of a matrix WITHOUT locality (bis_spmv.hip: launch_colslab).
//
// A matrix whose rows reach all over x (config 5's `unstr:80,80,80` as generated: a mesh numbered at random) gathers x
// through the fabric: x (12.3 MB) does not fit the 4 MB L2 of an XCD, every 8-byte operand costs a 64-byte fetch, and the
// one-pass CRS kernel runs at 0.9 TB/s of algorithmic bytes (1.39 ms; tools/colblock_probe.py).  Cut into K column slabs
// A = [A_0 | A_1 | ... | A_{K-1}] whose x slices fit the L2, the K passes  y = A_0 x ; y = y + A_1 x ; ...  gather from the L2
// (K = 6: 0.53 ms in all).  With the columns of every row ascending, the entries of slab k+1 follow those of slab k in the
// row, and pass k+1 continues the row's sum where pass k left it: the sum is the reference's left-to-right chain
// (kernels.hpp:25-39), bit for bit.  A row along which the slab index falls refuses the plan (the order inside a slab is free).
//
// Here: the K slab matrices (CRS copies of the column ranges: 12 bytes per non-zero once more) built on the device.
#include "bis_internal.hpp"

#include <rocprim/rocprim.hpp>

#include <algorithm>

namespace {

struct SlabPtrs {
    int32_t *col;
    double *val;
    const int32_t *rp;
};

// cnt[t * n + r] = entries of row r in slab t; flags[0] |= 1 where the slab index falls along a row.  A wave per row.
__global__ __launch_bounds__(256) void slab_count_kernel(const int32_t *__restrict__ rp, const int32_t *__restrict__ col, int64_t n,
                                                         int width, int K, int32_t *__restrict__ cnt, int *flags) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return; // (wave-uniform)
    const int64_t e = rp[r + 1];
    int mine = 0, prev_last = -1;
    bool unsorted = false;
    for (int64_t base = rp[r]; base < e; base += 64) {
        const bool in = base + lane < e;
        const int c = in ? col[base + lane] : 0;
        const int s = in ? min(K - 1, c / width) : 0x7fffffff;
        // what the passes need is that the SLAB index never falls along the row (ascending columns are the usual reason): the
        // entries of slab k+1 then follow those of slab k, whatever their order inside a slab
        int before = __shfl_up(s, 1);
        if (lane == 0) before = prev_last;
        unsorted |= in && s < before;
        const int left = (int)min((int64_t)64, e - base);
        prev_last = __shfl(s, left - 1);
        for (int t = 0; t < K; ++t) {
            const int m = __popcll(__ballot(in && s == t));
            if (lane == t) mine += m;
        }
    }
    if (lane < K) cnt[(size_t)lane * n + r] = mine;
    // (one flag for the whole matrix: a matrix whose rows are ALL out of order would queue 1.5 M atomics on it -- 15 ms)
    if (__ballot(unsorted) && lane == 0 && __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(flags, 1);
}

__global__ __launch_bounds__(256) void slab_fill_kernel(const int32_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                        const double *__restrict__ val, int64_t n, int width, int K,
                                                        const SlabPtrs *__restrict__ tab) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    int my_pos = lane < K ? tab[lane].rp[r] : 0; // lane t: where slab t's next entry of this row goes
    const int64_t e = rp[r + 1];
    for (int64_t base = rp[r]; base < e; base += 64) {
        const bool in = base + lane < e;
        const int c = in ? col[base + lane] : 0;
        const double v = in ? val[base + lane] : 0.0;
        const int s = in ? min(K - 1, c / width) : -1;
        for (int t = 0; t < K; ++t) {
            const unsigned long long m = __ballot(s == t);
            if (!m) continue;
            const int at = __shfl(my_pos, t);
            if (s == t) {
                const int q = at + __popcll(m & below);
                tab[t].col[q] = c;
                tab[t].val[q] = v;
            }
            if (lane == t) my_pos += __popcll(m);
        }
    }
}

} // namespace

void bis_spmv_colslab_free(bis_ctx *ctx, std::vector<bis_mat *> &slabs) {
    for (bis_mat *B : slabs) bis_mat_destroy(ctx, B);
    slabs.clear();
}

// slabs: K matrices of A's shape holding A's entries of the column ranges [t * width, (t + 1) * width) (the last one to the
// end), rows and the order inside a row kept.  *ok = false (and no slabs): the rows are not ascending.
bis_status bis_spmv_colslab_build(bis_ctx *ctx, const bis_mat *A, int K, std::vector<bis_mat *> &slabs, bool *ok) {
    *ok = false;
    slabs.clear();
    const int64_t n = A->n_rows;
    if (A->rp64 || K < 2 || K > 32 || n == 0 || A->nnz == 0 || A->n_cols >= ((int64_t)1 << 31)) return BIS_OK;
    const int width = (int)((A->n_cols + K - 1) / K);
    int32_t *cnt = nullptr, *rps = nullptr;
    int *flags = nullptr;
    void *tmp = nullptr;
    SlabPtrs *tab = nullptr;
    auto cleanup = [&](bis_status rc) {
        hipFree(cnt); hipFree(rps); hipFree(flags); hipFree(tmp); hipFree(tab);
        if (rc != BIS_OK) bis_spmv_colslab_free(ctx, slabs);
        return rc;
    };
#define BIS_SL_CHECK(call)                                                                                             \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return cleanup(BIS_ERR_HIP); } \
    } while (0)
    BIS_SL_CHECK(hipMalloc(&cnt, sizeof(int32_t) * (size_t)K * (size_t)n));
    BIS_SL_CHECK(hipMalloc(&rps, sizeof(int32_t) * (size_t)K * (size_t)(n + 1)));
    BIS_SL_CHECK(hipMalloc(&flags, sizeof(int) * 4));
    BIS_SL_CHECK(hipMemsetAsync(flags, 0, sizeof(int) * 4, ctx->stream));
    const unsigned grid = (unsigned)((n + 3) / 4);
    hipLaunchKernelGGL(slab_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col, n, width, K, cnt, flags);
    int h = 0;
    BIS_SL_CHECK(hipMemcpyAsync(&h, flags, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    BIS_SL_CHECK(hipStreamSynchronize(ctx->stream));
    if (h) return cleanup(BIS_OK); // a row that returns to an earlier slab: the passes would reorder its sum
    size_t tmp_bytes = 0;
    BIS_SL_CHECK(rocprim::inclusive_scan(nullptr, tmp_bytes, cnt, rps + 1, (size_t)n, rocprim::plus<int32_t>(), ctx->stream));
    BIS_SL_CHECK(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)));
    for (int t = 0; t < K; ++t) {
        int32_t *rp_t = rps + (size_t)t * (size_t)(n + 1);
        BIS_SL_CHECK(hipMemsetAsync(rp_t, 0, sizeof(int32_t), ctx->stream));
        BIS_SL_CHECK(rocprim::inclusive_scan(tmp, tmp_bytes, cnt + (size_t)t * (size_t)n, rp_t + 1, (size_t)n, rocprim::plus<int32_t>(), ctx->stream));
    }
    std::vector<int32_t> totals((size_t)K, 0);
    for (int t = 0; t < K; ++t)
        BIS_SL_CHECK(hipMemcpyAsync(&totals[(size_t)t], rps + (size_t)t * (size_t)(n + 1) + n, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    BIS_SL_CHECK(hipStreamSynchronize(ctx->stream));
    int64_t sum = 0;
    for (int t = 0; t < K; ++t) sum += totals[(size_t)t];
    if (sum != A->nnz) { ctx->err = "bis_spmv_colslab_build: slab counts do not add up (internal)"; return cleanup(BIS_ERR_INVALID); }
    std::vector<SlabPtrs> host_tab((size_t)K);
    for (int t = 0; t < K; ++t) {
        bis_mat *B = nullptr;
        if (bis_status st = bis_mat_alloc(ctx, n, A->n_cols, totals[(size_t)t], false, &B)) return cleanup(st);
        slabs.push_back(B);
        BIS_SL_CHECK(hipMemcpyAsync(B->row_ptr, rps + (size_t)t * (size_t)(n + 1), sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToDevice, ctx->stream));
        host_tab[(size_t)t] = SlabPtrs{B->col, B->val, (const int32_t *)B->row_ptr};
    }
    BIS_SL_CHECK(hipMalloc(&tab, sizeof(SlabPtrs) * (size_t)K));
    BIS_SL_CHECK(hipMemcpyAsync(tab, host_tab.data(), sizeof(SlabPtrs) * (size_t)K, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(slab_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col, A->val, n, width, K, tab);
    BIS_SL_CHECK(hipGetLastError());
    BIS_SL_CHECK(hipStreamSynchronize(ctx->stream)); // (host_tab / tab are read by the copy and the kernel)
#undef BIS_SL_CHECK
    for (bis_mat *B : slabs) {
        if (bis_status st = bis_mat_finalize(ctx, B)) return cleanup(st);
        B->vd_state = -1; // arbitrary values: no dictionary scan per slab
    }
    *ok = true;
    return cleanup(BIS_OK);
}
