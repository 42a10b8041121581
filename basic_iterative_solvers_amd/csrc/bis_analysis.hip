// bis_analysis.hip -- dependency-level analysis of a strictly triangular matrix
// on the device (setup of bis_sptrsv / bis_bsptrsv / bis_mat_ilu0; the host
// version in bis_sptrsv.hip downloads the pattern and walks it serially --
// 0.4-0.5 s per triangle of HPCG-256).
//
//   level[r] = 1 + max level of the rows r depends on (0 without dependencies),
//   computed in ONE launch: a persistent grid takes 256-row tickets in the
//   order of the substitution (ascending rows for a lower, descending for an
//   upper triangle), every lane polls the levels of its dependencies (a
//   sentinel marks "not yet known"; the published level is the flag) -- the
//   wait loop of the triangular solve itself (bounded, publishing store
//   predicated inside volatile asm, see bis_sptrsv.hip).  The same pass checks
//   that every entry lies strictly on the required side of the diagonal.
//   perm  = rows sorted by level, ascending rows inside a level: stable radix
//           sort of (level, row) pairs (rocPRIM);
//   level_ptr from the boundaries of the sorted keys.
#include <cstring>

#include "bis_internal.hpp"

#include <rocprim/rocprim.hpp>

#include <algorithm>

namespace {

constexpr unsigned kLvSpinLimit = 1u << 22;
constexpr int kLvBatch = 8;

template <typename RP, bool BACKWARD>
__global__ __launch_bounds__(256) void level_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                    int64_t n, int *level /* -1 = not yet */, unsigned *ticket,
                                                    int *status /* [0] not triangular, [1] lost hand-off, [2] max level */) {
    __shared__ unsigned s_ticket;
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        const unsigned t = s_ticket;
        __syncthreads();
        const int64_t base = (int64_t)t * 256;
        if (base >= n) return;
        const int64_t i = base + threadIdx.x;
        const bool valid = i < n; // lanes past the end run an empty row (no divergent branch at the loop tail)
        {
            const int64_t r = valid ? (BACKWARD ? n - 1 - i : i) : 0;
            int64_t k = valid ? (int64_t)row_ptr[r] : 0;
            const int64_t e = valid ? (int64_t)row_ptr[r + 1] : 0;
            // entries are taken towards the diagonal (an upper row from its last entry down): with ascending columns the row that
            // was finished last -- the one this row waits for -- is then in the LAST batch, and the trips to memory of the earlier
            // batches overlap that wait instead of following it (RCM-ordered 1.5 M rows, 7119 levels: 97 -> 2x ms backward)
            const int64_t mirror = k + e - 1;
            int lvl = 0;
            bool bad = false, lost = false, done = false;
            unsigned spins = 0;
            int lv[kLvBatch], pc[kLvBatch];
#pragma unroll
            for (int j = 0; j < kLvBatch; ++j) { lv[j] = 0; pc[j] = 0; }
            int ready = 0, in_batch = 0;
            while (!done) { // one loop, bounded work per trip (lanes of a wave wait for each other)
                bool publish = false;
                if (ready == in_batch) {
                    k += in_batch;
                    if (k == e) {
                        publish = true;
                        in_batch = ready = 0;
                    } else {
                        in_batch = e - k < (int64_t)kLvBatch ? (int)(e - k) : kLvBatch;
#pragma unroll
                        for (int j = 0; j < kLvBatch; ++j) {
                            pc[j] = j < in_batch ? col[BACKWARD ? mirror - (k + j) : k + j] : 0;
                            if (j < in_batch) {
                                const bool side = BACKWARD ? pc[j] > r : pc[j] < r;
                                if (pc[j] < 0 || pc[j] >= n || !side) { bad = true; pc[j] = -1; }
                            }
                        }
#pragma unroll
                        for (int j = 0; j < kLvBatch; ++j) // an offending entry is reported, not waited for
                            lv[j] = (j < in_batch && pc[j] >= 0) ? __hip_atomic_load(&level[pc[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                                 : 0;
                        ready = 0;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < kLvBatch; ++j)
                        if (j >= ready && j < in_batch && lv[j] < 0)
                            lv[j] = __hip_atomic_load(&level[pc[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < kLvBatch; ++j) {
                    if (ready == j && j < in_batch && lv[j] >= 0) {
                        if (pc[j] >= 0) lvl = max(lvl, lv[j] + 1);
                        ready = j + 1;
                    }
                }
                if (ready < in_batch) {
                    if (++spins > kLvSpinLimit) { publish = true; lost = true; }
                    else __builtin_amdgcn_s_sleep(1);
                }
                {
                    int *dst = &level[r];
                    const unsigned pflag = (publish && valid) ? 1u : 0u;
                    unsigned long long saved_exec;
                    asm volatile("v_cmp_ne_u32_e32 vcc, 0, %3\n\ts_and_saveexec_b64 %0, vcc\n\t"
                                 "global_store_dword %1, %2, off sc1\n\ts_mov_b64 exec, %0"
                                 : "=&s"(saved_exec) : "v"(dst), "v"(lvl), "v"(pflag) : "vcc", "memory");
                }
                if (publish) done = true;
            }
            // error flags behind wave-uniform conditions only: a divergent branch at the tail of
            // the ticket loop would let some lanes run ahead into the next trip's barrier
            if (__any(bad)) atomicOr(&status[0], bad ? 1 : 0);
            if (__any(lost)) atomicOr(&status[1], lost ? 1 : 0);
            atomicMax(&status[2], lvl);
        }
    }
}

__global__ __launch_bounds__(256) void iota_fill_kernel(int *level, int32_t *rows, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { level[i] = -1; rows[i] = (int32_t)i; }
}

// level_ptr[l] = first position of level l in the sorted order; level_ptr[n_levels] = n
__global__ __launch_bounds__(256) void level_bounds_kernel(const int *__restrict__ keys, int64_t n, int n_levels,
                                                           int64_t *__restrict__ level_ptr) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (i == 0 || keys[i] != keys[i - 1]) level_ptr[keys[i]] = i;
        if (i == n - 1) level_ptr[n_levels] = n;
    }
}

} // namespace

// ---------------------------------------------------------------------------
// Contiguous independent row blocks (the schedule of a colour-sorted matrix).
// Forward: block [b_k, b_{k+1}) may be swept by one streaming launch iff every
// dependency of its rows lies before b_k, i.e. b_{k+1} = first r >= b_k whose
// largest column is >= b_k.  Backward: mirrored from the end.  This needs neither
// exact dependency levels nor a row list; a matrix that does not decompose into
// at most max_blocks such blocks (any natural ordering) is reported as such.
// ---------------------------------------------------------------------------
namespace {

template <typename RP, bool BACKWARD>
__global__ __launch_bounds__(256) void row_extreme_col_kernel(const RP *__restrict__ row_ptr,
                                                              const int32_t *__restrict__ col, int64_t n,
                                                              int32_t *__restrict__ m, int32_t *__restrict__ iota,
                                                              int *__restrict__ bad) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    int lo = INT32_MAX, hi = -1;
    for (int64_t k = (int64_t)row_ptr[r]; k < (int64_t)row_ptr[r + 1]; ++k) { lo = min(lo, col[k]); hi = max(hi, col[k]); }
    // strictly triangular, columns in range
    if (hi >= 0 && (BACKWARD ? (lo <= r || hi >= n) : (hi >= r || lo < 0))) atomicOr(bad, 1);
    m[r] = BACKWARD ? lo : hi;
    if (iota) iota[r] = (int32_t)r;
}

// forward: out = min { r in [start, n) : m[r] >= start }  (n if none)
// backward: out = max { r in [0, end) : m[r] < end }       (-1 if none)
template <bool BACKWARD>
__global__ __launch_bounds__(256) void block_boundary_kernel(const int32_t *__restrict__ m, int64_t n, int64_t edge,
                                                             long long *out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    long long best = BACKWARD ? -1 : (long long)n;
    if (BACKWARD) {
        for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < edge; r += stride)
            if (m[r] < edge) best = r > best ? r : best;
    } else {
        for (int64_t r = edge + (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride)
            if (m[r] >= edge) { best = r; break; } // ascending per lane: the first hit is this lane's minimum
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const long long o = __shfl_down(best, off, 64);
        best = BACKWARD ? (o > best ? o : best) : (o < best ? o : best);
    }
    if ((threadIdx.x & 63) == 0) {
        if (BACKWARD) atomicMax(out, best); else atomicMin(out, best);
    }
}

} // namespace

// bounds: block edges in processing order (forward: b_0=0 < b_1 < ... = n; backward: n = e_0 > e_1 > ... = 0);
// empty if the matrix needs more than max_blocks blocks.  perm_dev (optional): filled with the identity --
// the "level-sorted" row list of a matrix whose blocks are its levels.
bis_status bis_trsv_blocks_device(bis_ctx *ctx, const bis_mat *T, bool backward, int max_blocks,
                                  std::vector<int64_t> &bounds, int32_t *perm_dev, bool &triangular) {
    bounds.clear();
    triangular = true;
    const int64_t n = T->n_rows;
    if (n == 0) return BIS_OK;
    int32_t *m = nullptr;
    long long *out = nullptr;
    auto cleanup = [&](bis_status rc) { hipFree(m); hipFree(out); return rc; };
#define BIS_BL_CHECK(call)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return cleanup(BIS_ERR_HIP); } \
    } while (0)
    BIS_BL_CHECK(hipMalloc(&m, sizeof(int32_t) * (size_t)n));
    BIS_BL_CHECK(hipMalloc(&out, sizeof(long long) * 2));
    int *bad = reinterpret_cast<int *>(out + 1);
    BIS_BL_CHECK(hipMemsetAsync(out, 0, sizeof(long long) * 2, ctx->stream));
    const unsigned n_blk = (unsigned)((n + 255) / 256);
    if (T->rp64) {
        if (backward) hipLaunchKernelGGL((row_extreme_col_kernel<int64_t, true>), dim3(n_blk), dim3(256), 0, ctx->stream, (const int64_t *)T->row_ptr, T->col, n, m, perm_dev, bad);
        else hipLaunchKernelGGL((row_extreme_col_kernel<int64_t, false>), dim3(n_blk), dim3(256), 0, ctx->stream, (const int64_t *)T->row_ptr, T->col, n, m, perm_dev, bad);
    } else {
        if (backward) hipLaunchKernelGGL((row_extreme_col_kernel<int32_t, true>), dim3(n_blk), dim3(256), 0, ctx->stream, (const int32_t *)T->row_ptr, T->col, n, m, perm_dev, bad);
        else hipLaunchKernelGGL((row_extreme_col_kernel<int32_t, false>), dim3(n_blk), dim3(256), 0, ctx->stream, (const int32_t *)T->row_ptr, T->col, n, m, perm_dev, bad);
    }
    {
        int hb = 0;
        BIS_BL_CHECK(hipMemcpyAsync(&hb, bad, sizeof hb, hipMemcpyDeviceToHost, ctx->stream));
        BIS_BL_CHECK(hipStreamSynchronize(ctx->stream));
        if (hb) { triangular = false; return cleanup(BIS_OK); }
    }
    std::vector<int64_t> b;
    int64_t edge = backward ? n : 0;
    b.push_back(edge);
    const unsigned grid = std::min<unsigned>(n_blk, 1024);
    while (true) {
        const long long init = backward ? -1 : (long long)n;
        BIS_BL_CHECK(hipMemcpyAsync(out, &init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
        if (backward) hipLaunchKernelGGL(block_boundary_kernel<true>, dim3(grid), dim3(256), 0, ctx->stream, m, n, edge, out);
        else hipLaunchKernelGGL(block_boundary_kernel<false>, dim3(grid), dim3(256), 0, ctx->stream, m, n, edge, out);
        long long h = 0;
        BIS_BL_CHECK(hipMemcpyAsync(&h, out, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        BIS_BL_CHECK(hipStreamSynchronize(ctx->stream));
        const int64_t next = backward ? (int64_t)h + 1 : (int64_t)h; // backward: the block starts right after the highest violating row
        b.push_back(next);
        edge = next;
        if (backward ? next <= 0 : next >= n) break;
        // not a colour-sorted matrix: too many blocks (no shortcut on block sizes -- the last colours of a
        // greedy colouring hold a handful of rows; a natural ordering costs max_blocks small reductions once)
        if ((int)b.size() - 1 >= max_blocks) return cleanup(BIS_OK);
    }
#undef BIS_BL_CHECK
    bounds = b;
    return cleanup(BIS_OK);
}

// perm_dev: n int32 (device).  triangular = false: an entry on the wrong side of the
// diagonal (or out of range) was found; the other outputs are then meaningless.
bis_status bis_trsv_analyse_device(bis_ctx *ctx, const bis_mat *T, bool backward, int32_t *perm_dev,
                                   std::vector<int64_t> &level_ptr, int &n_levels, int64_t &max_width,
                                   bool &triangular, int **level_out) {
    const int64_t n = T->n_rows;
    if (level_out) *level_out = nullptr;
    triangular = true;
    n_levels = 0;
    max_width = 0;
    level_ptr.assign(1, 0);
    if (n == 0) return BIS_OK;
    int *level = nullptr, *keys_out = nullptr, *status = nullptr;
    int32_t *rows = nullptr;
    unsigned *ticket = nullptr;
    int64_t *lp_dev = nullptr;
    void *tmp = nullptr;
    auto cleanup = [&](bis_status rc) {
        hipFree(level); hipFree(keys_out); hipFree(status); hipFree(rows); hipFree(ticket); hipFree(lp_dev); hipFree(tmp);
        return rc;
    };
#define BIS_AN_CHECK(call)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return cleanup(BIS_ERR_HIP); } \
    } while (0)
    BIS_AN_CHECK(hipMalloc(&level, sizeof(int) * (size_t)n));
    BIS_AN_CHECK(hipMalloc(&keys_out, sizeof(int) * (size_t)n));
    BIS_AN_CHECK(hipMalloc(&rows, sizeof(int32_t) * (size_t)n));
    BIS_AN_CHECK(hipMalloc(&status, sizeof(int) * 4));
    BIS_AN_CHECK(hipMalloc(&ticket, sizeof(unsigned) * 4));
    BIS_AN_CHECK(hipMemsetAsync(status, 0, sizeof(int) * 4, ctx->stream));
    BIS_AN_CHECK(hipMemsetAsync(ticket, 0, sizeof(unsigned) * 4, ctx->stream));
    const int n_blk = (int)((n + 255) / 256);
    hipLaunchKernelGGL(iota_fill_kernel, dim3((unsigned)std::min(n_blk, 2048)), dim3(256), 0, ctx->stream, level, rows, n);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_blk, (int64_t)ctx->n_cus * 4)); // resident by construction
#define BIS_AN_LEVEL(RP, BW)                                                                      \
    hipLaunchKernelGGL((level_kernel<RP, BW>), dim3(grid), dim3(256), 0, ctx->stream, (const RP *)T->row_ptr, T->col, n, \
                       level, ticket, status)
    if (T->rp64) { if (backward) BIS_AN_LEVEL(int64_t, true); else BIS_AN_LEVEL(int64_t, false); }
    else { if (backward) BIS_AN_LEVEL(int32_t, true); else BIS_AN_LEVEL(int32_t, false); }
#undef BIS_AN_LEVEL
    int h[4] = {0, 0, 0, 0};
    BIS_AN_CHECK(hipMemcpyAsync(h, status, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    BIS_AN_CHECK(hipStreamSynchronize(ctx->stream));
    if (h[0]) { triangular = false; return cleanup(BIS_OK); }
    if (h[1]) { ctx->err = "sptrsv analysis: a hand-off was lost"; return cleanup(BIS_ERR_HIP); }
    n_levels = h[2] + 1;
    int bits = 1;
    while ((1ll << bits) < n_levels) ++bits;
    size_t tmp_bytes = 0;
    BIS_AN_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, level, keys_out, rows, perm_dev, (size_t)n, 0, bits, ctx->stream));
    BIS_AN_CHECK(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)));
    BIS_AN_CHECK(rocprim::radix_sort_pairs(tmp, tmp_bytes, level, keys_out, rows, perm_dev, (size_t)n, 0, bits, ctx->stream));
    BIS_AN_CHECK(hipMalloc(&lp_dev, sizeof(int64_t) * (size_t)(n_levels + 1)));
    hipLaunchKernelGGL(level_bounds_kernel, dim3((unsigned)std::min(n_blk, 2048)), dim3(256), 0, ctx->stream, keys_out, n,
                       n_levels, lp_dev);
    level_ptr.assign((size_t)n_levels + 1, 0);
    BIS_AN_CHECK(hipMemcpyAsync(level_ptr.data(), lp_dev, sizeof(int64_t) * (size_t)(n_levels + 1), hipMemcpyDeviceToHost,
                                ctx->stream));
    BIS_AN_CHECK(hipStreamSynchronize(ctx->stream));
#undef BIS_AN_CHECK
    for (int l = 0; l < n_levels; ++l) max_width = std::max<int64_t>(max_width, level_ptr[l + 1] - level_ptr[l]);
    if (level_out) { *level_out = level; level = nullptr; } // (the sort read it, it did not change it)
    return cleanup(BIS_OK);
}
