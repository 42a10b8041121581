// bis_order.hip -- breadth-first and reverse Cuthill-McKee orderings on the device, and B = P A P^T for
// any permutation (the roles of SMAX's PERM_MODE BFS / RCM and of its permute_mat in the reference:
// CMakeLists.txt:128-133, utilities/smax_helpers.hpp:44-80).
//
// The orderings are defined by the sequential queue algorithm of the host version
// (host/utilities/permute.hpp bfs_like_permutation): components are started from the lowest-numbered
// unseen vertex (BFS) or the unseen vertex of smallest (degree, index) (RCM); a vertex popped from the
// queue appends its unseen neighbours in ascending index order (BFS) or ascending (degree, index) order
// (RCM); RCM reverses the final order.  The queue order is reproduced level by level:
//   * the children of level l are the unseen neighbours of its vertices; a child belongs to the parent
//     that was popped first, i.e. the adjacent level-l vertex of smallest position (atomicMin),
//   * inside level l+1 the queue order is (position of the parent, key of the child) with key = index
//     or rank in (degree, index) order: one stable radix sort of 64-bit keys per level.
// The result is the host algorithm's permutation entry for entry (tests compare the permutation files).
// Structurally symmetric patterns only (the adjacency is then the matrix' own rows); anything else, and
// graphs of more than 64 components, are left to the host version (BIS_ERR_UNSUPPORTED).
#include "bis_internal.hpp"

#include <rocprim/rocprim.hpp>

#include <algorithm>

namespace {

// status[0] |= 1 if some entry (r,c), c != r, has no mirror (c,r), or a row holds a column twice; deg[r] = entries off the diagonal.
//
// One WAVE per row.  Only the entries BELOW the diagonal are looked up (all 64 lanes read the mirrored row's columns at once: one
// coalesced request per 64 entries instead of a lane walking the row alone, four entries' requests in flight), and the rows
// count their entries below and above the diagonal into status[6..7] (one 64-bit balance): without repeated entries the mirrors of
// the lower entries are distinct upper entries, so "every lower entry has its mirror" + "as many upper as lower entries" is
// structural symmetry.  unstr:80,80,80 (1.04e8 entries, no locality): 69 ms with a lane per row and every entry looked up.
template <typename RP>
__global__ __launch_bounds__(256) void symmetry_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col, int64_t n,
                                                       int *__restrict__ deg, int *status) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return; // (wave-uniform)
    const int64_t s = (int64_t)row_ptr[r], e = (int64_t)row_ptr[r + 1];
    bool bad = false;
    int d = 0;
    long long balance = 0;
    for (int64_t base = s; base < e; base += 64) {
        const int cnt = e - base < 64 ? (int)(e - base) : 64;
        const int c_mine = lane < cnt ? col[base + lane] : -1;
        const bool lower = c_mine >= 0 && (int64_t)c_mine < r;
        d += __popcll(__ballot(c_mine >= 0 && (int64_t)c_mine != r));
        balance += (long long)__popcll(__ballot(lower)) - (long long)__popcll(__ballot((int64_t)c_mine > r));
        // a column twice in the row: among the entries of this chunk, and against the chunks before it
        for (int j = 1; j < cnt; ++j) {
            const int c = __shfl(c_mine, j);
            bad |= (__ballot(c_mine == c) & ((1ull << j) - 1ull)) != 0ull;
        }
        for (int64_t q = s; q < base; q += 64) {
            const int w = col[q + lane];
            for (int j = 0; j < cnt; ++j) bad |= __ballot(w == __shfl(c_mine, j)) != 0ull;
        }
        // mirrors of the lower entries
        int64_t qs = 0;
        int ql = 0;
        if (lower) { qs = (int64_t)row_ptr[c_mine]; ql = (int)((int64_t)row_ptr[c_mine + 1] - qs); }
        const unsigned long long todo = __ballot(lower);
        for (int j0 = 0; j0 < cnt; j0 += 4) {
            if (((todo >> j0) & 0xFull) == 0ull) continue;
            int v[4], len[4];
            int64_t start[4];
            bool on[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = (j0 + u) & 63;
                on[u] = j0 + u < 64 && ((todo >> j) & 1ull) != 0ull;
                start[u] = __shfl(qs, j);
                len[u] = on[u] ? __shfl(ql, j) : 0;
                v[u] = lane < len[u] ? col[start[u] + lane] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!on[u]) continue;
                bool found = __ballot((int64_t)v[u] == r) != 0ull;
                for (int q = 64; q < len[u] && !found; q += 64) {
                    const int w = q + lane < len[u] ? col[start[u] + q + lane] : -1;
                    found = __ballot((int64_t)w == r) != 0ull;
                }
                bad |= !found;
            }
        }
    }
    if (lane == 0) {
        deg[r] = d;
        if (bad && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(status, 1);
        if (balance != 0) atomicAdd((unsigned long long *)(status + 6), (unsigned long long)balance);
    }
}

__global__ __launch_bounds__(256) void degree_keys_kernel(const int *__restrict__ deg, int64_t n, unsigned long long *keys, int32_t *ids) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < n) { keys[v] = ((unsigned long long)(unsigned)deg[v] << 32) | (unsigned)v; ids[v] = (int32_t)v; }
}
__global__ __launch_bounds__(256) void iota_kernel(int32_t *p, int64_t n) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < n) p[v] = (int32_t)v;
}
__global__ __launch_bounds__(256) void rank_kernel(const int32_t *__restrict__ by_key, int64_t n, int32_t *rank) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) rank[by_key[i]] = (int32_t)i;
}
__global__ __launch_bounds__(256) void fill32_kernel(int32_t *p, int64_t n, int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// first unseen vertex in key order at or after `cursor`: out[0] = min i with pos[by_key[i]] < 0
__global__ __launch_bounds__(256) void first_unseen_kernel(const int32_t *__restrict__ by_key, const int32_t *__restrict__ pos,
                                                           int64_t cursor, int64_t n, int *out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int best = INT32_MAX;
    for (int64_t i = cursor + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        if (pos[by_key[i]] < 0) { best = (int)i; break; }
    for (int off = 32; off > 0; off >>= 1) best = min(best, __shfl_down(best, off, 64));
    if ((threadIdx.x & 63) == 0 && best != INT32_MAX) atomicMin(out, best);
}

__global__ void start_component_kernel(const int32_t *by_key, const int *first, int32_t *order, int32_t *pos, int64_t count) {
    const int32_t v = by_key[*first];
    order[count] = v;
    pos[v] = (int32_t)count;
}

// expand the frontier order[f0, f1): claim unseen neighbours, parent = smallest frontier position
template <typename RP>
__global__ __launch_bounds__(256) void expand_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                     const int32_t *__restrict__ order, int64_t f0, int64_t f1,
                                                     const int32_t *__restrict__ pos, int32_t *parent, int32_t *mark, int32_t level_id,
                                                     int32_t *next_list, int *next_count) {
    const int64_t p = f0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= f1) return;
    const int64_t v = order[p];
    for (int64_t k = (int64_t)row_ptr[v]; k < (int64_t)row_ptr[v + 1]; ++k) {
        const int32_t w = col[k];
        if (w == v || pos[w] >= 0) continue;
        atomicMin(&parent[w], (int32_t)p);
        if (atomicExch(&mark[w], level_id) != level_id) next_list[atomicAdd(next_count, 1)] = w;
    }
}

__global__ __launch_bounds__(256) void child_keys_kernel(const int32_t *__restrict__ next_list, int m, const int32_t *__restrict__ parent,
                                                         const int32_t *__restrict__ rank, unsigned long long *keys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) { const int32_t w = next_list[i]; keys[i] = ((unsigned long long)(unsigned)parent[w] << 32) | (unsigned)rank[w]; }
}

__global__ __launch_bounds__(256) void append_kernel(const int32_t *__restrict__ sorted, int m, int64_t count, int32_t *order, int32_t *pos) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) { const int32_t w = sorted[i]; order[count + i] = w; pos[w] = (int32_t)(count + i); }
}

__global__ __launch_bounds__(256) void reverse_kernel(const int32_t *__restrict__ in, int64_t n, int32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[n - 1 - i];
}

// ---- B = P A P^T -------------------------------------------------------------------------------------
template <typename RP>
__global__ __launch_bounds__(256) void pm_count_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ perm, int64_t n,
                                                       int64_t *__restrict__ blk) {
    __shared__ double lds[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int len = 0;
    if (i < n) { const int o = perm[i]; len = (int)(row_ptr[o + 1] - row_ptr[o]); }
    const double tot = block_sum<256>((double)len, lds);
    if (threadIdx.x == 0) blk[blockIdx.x] = (int64_t)tot;
}
__global__ __launch_bounds__(256) void pm_scan_kernel(int64_t *blk, int n_blk) {
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
__global__ __launch_bounds__(256) void pm_fill_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                      const double *__restrict__ val, const int32_t *__restrict__ perm,
                                                      const int32_t *__restrict__ inv, int64_t n, const int64_t *__restrict__ blk,
                                                      RP *__restrict__ rpB, int32_t *__restrict__ colB, double *__restrict__ valB) {
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
    __shared__ int64_t sa[256];
    __shared__ int sn[256];
    {
        const int64_t p = blk[blockIdx.x] + sc[threadIdx.x] - len;
        __syncthreads();
        sc[threadIdx.x] = p; // (from here on: where the row's entries go)
        sa[threadIdx.x] = a;
        sn[threadIdx.x] = len;
        if (i < n) {
            rpB[i] = (RP)p;
            if (i == n - 1) rpB[n] = (RP)(p + len);
        }
        __syncthreads();
    }
    // the entries, a WAVE per row (coalesced on both sides; entries keep their order inside the row)
    const int lane = threadIdx.x & 63;
    for (int t = threadIdx.x >> 6; t < 256; t += 4) {
        if ((int64_t)blockIdx.x * 256 + t >= n) break; // (wave-uniform)
        const int64_t src = sa[t], dst = sc[t];
        const int ln = sn[t];
        for (int q = lane; q < ln; q += 64) {
            colB[dst + q] = inv[col[src + q]];
            valB[dst + q] = val[src + q];
        }
    }
}
__global__ __launch_bounds__(256) void invert_kernel(const int32_t *__restrict__ perm, int64_t n, int32_t *__restrict__ inv, int *status) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t o = perm[i];
    if (o < 0 || o >= n) { atomicOr(status, 1); return; }
    if (atomicExch(&inv[o], (int32_t)i) != -1) atomicOr(status, 1); // not a permutation
}

#define BIS_OR_CHECK(call)                                                                                              \
    do {                                                                                                                \
        hipError_t e_ = (call);                                                                                         \
        if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return cleanup(BIS_ERR_HIP); } \
    } while (0)

template <typename RP>
bis_status permute_t(bis_ctx *ctx, const bis_mat *A, const int32_t *perm_dev, bis_mat **B_out) {
    const int64_t n = A->n_rows;
    const RP *rp = (const RP *)A->row_ptr;
    const int n_blk = (int)((n + 255) / 256);
    int32_t *inv = nullptr;
    int64_t *blk = nullptr;
    int *status = nullptr;
    bis_mat *B = nullptr;
    auto cleanup = [&](bis_status rc) {
        hipFree(inv); hipFree(blk); hipFree(status);
        if (rc != BIS_OK && B) bis_mat_destroy(ctx, B);
        return rc;
    };
    BIS_OR_CHECK(hipMalloc(&inv, sizeof(int32_t) * (size_t)std::max<int64_t>(n, 1)));
    BIS_OR_CHECK(hipMalloc(&blk, sizeof(int64_t) * (size_t)(n_blk + 1)));
    BIS_OR_CHECK(hipMalloc(&status, sizeof(int) * 4));
    BIS_OR_CHECK(hipMemsetAsync(status, 0, sizeof(int) * 4, ctx->stream));
    if (n_blk) {
        hipLaunchKernelGGL(fill32_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, inv, n, (int32_t)-1);
        hipLaunchKernelGGL(invert_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, perm_dev, n, inv, status);
    }
    int h = 0;
    BIS_OR_CHECK(hipMemcpyAsync(&h, status, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    BIS_OR_CHECK(hipStreamSynchronize(ctx->stream));
    if (h) { ctx->err = "bis_mat_permute: perm is not a permutation of 0..n-1"; return cleanup(BIS_ERR_INVALID); }
    bis_status st = bis_mat_alloc(ctx, n, A->n_cols, A->nnz, A->rp64, &B);
    if (st != BIS_OK) return cleanup(st);
    if (n_blk) {
        hipLaunchKernelGGL(pm_count_kernel<RP>, dim3(n_blk), dim3(256), 0, ctx->stream, rp, perm_dev, n, blk);
        hipLaunchKernelGGL(pm_scan_kernel, dim3(1), dim3(256), 0, ctx->stream, blk, n_blk);
        hipLaunchKernelGGL(pm_fill_kernel<RP>, dim3(n_blk), dim3(256), 0, ctx->stream, rp, A->col, A->val, perm_dev, inv, n, blk,
                           (RP *)B->row_ptr, B->col, B->val);
    } else {
        BIS_OR_CHECK(hipMemsetAsync(B->row_ptr, 0, A->rp64 ? 8 : 4, ctx->stream));
    }
    BIS_OR_CHECK(hipGetLastError());
    BIS_OR_CHECK(hipStreamSynchronize(ctx->stream));
    st = bis_mat_finalize(ctx, B);
    if (st != BIS_OK) return cleanup(st);
    *B_out = B;
    return cleanup(BIS_OK);
}

template <typename RP>
bis_status bfs_order_t(bis_ctx *ctx, const bis_mat *A, bool rcm, int32_t *perm_dev) {
    const int64_t n = A->n_rows;
    const RP *rp = (const RP *)A->row_ptr;
    const unsigned n_blk = (unsigned)((n + 255) / 256);
    int *deg = nullptr, *status = nullptr;
    int32_t *by_key = nullptr, *rank = nullptr, *pos = nullptr, *parent = nullptr, *mark = nullptr, *order = nullptr, *next_list = nullptr,
            *sorted = nullptr, *ids = nullptr;
    unsigned long long *keys = nullptr, *keys_out = nullptr;
    void *tmp = nullptr;
    size_t tmp_cap = 0;
    auto cleanup = [&](bis_status rc) {
        hipFree(deg); hipFree(status); hipFree(by_key); hipFree(rank); hipFree(pos); hipFree(parent); hipFree(mark); hipFree(order);
        hipFree(next_list); hipFree(sorted); hipFree(ids); hipFree(keys); hipFree(keys_out); hipFree(tmp);
        return rc;
    };
    const size_t n1 = (size_t)std::max<int64_t>(n, 1);
    BIS_OR_CHECK(hipMalloc(&deg, 4 * n1));
    BIS_OR_CHECK(hipMalloc(&status, sizeof(int) * 8));
    BIS_OR_CHECK(hipMalloc(&by_key, 4 * n1));
    BIS_OR_CHECK(hipMalloc(&rank, 4 * n1));
    BIS_OR_CHECK(hipMalloc(&pos, 4 * n1));
    BIS_OR_CHECK(hipMalloc(&parent, 4 * n1));
    BIS_OR_CHECK(hipMalloc(&mark, 4 * n1));
    BIS_OR_CHECK(hipMalloc(&order, 4 * n1));
    BIS_OR_CHECK(hipMalloc(&next_list, 4 * n1));
    BIS_OR_CHECK(hipMalloc(&sorted, 4 * n1));
    BIS_OR_CHECK(hipMalloc(&ids, 4 * n1));
    BIS_OR_CHECK(hipMalloc(&keys, 8 * n1));
    BIS_OR_CHECK(hipMalloc(&keys_out, 8 * n1));
    BIS_OR_CHECK(hipMemsetAsync(status, 0, sizeof(int) * 8, ctx->stream));
    if (n == 0) return cleanup(BIS_OK);
    hipLaunchKernelGGL(symmetry_kernel<RP>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, rp, A->col, n, deg, status);
    int hs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    BIS_OR_CHECK(hipMemcpyAsync(hs, status, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
    BIS_OR_CHECK(hipStreamSynchronize(ctx->stream));
    const int h = hs[0] | (hs[6] != 0) | (hs[7] != 0); // [6..7]: entries below minus entries above the diagonal
    if (h) { ctx->err = "bis_mat_bfs_order: pattern is not structurally symmetric (or has repeated entries): use the host ordering"; return cleanup(BIS_ERR_UNSUPPORTED); }
    auto sort_pairs = [&](unsigned long long *k_in, unsigned long long *k_out, int32_t *v_in, int32_t *v_out, size_t m) -> hipError_t {
        size_t bytes = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, k_in, k_out, v_in, v_out, m, 0, 64, ctx->stream);
        if (e != hipSuccess) return e;
        if (bytes > tmp_cap) { hipFree(tmp); tmp = nullptr; e = hipMalloc(&tmp, bytes); if (e != hipSuccess) return e; tmp_cap = bytes; }
        return rocprim::radix_sort_pairs(tmp, bytes, k_in, k_out, v_in, v_out, m, 0, 64, ctx->stream);
    };
    if (rcm) { // candidates and children in ascending (degree, index) order
        hipLaunchKernelGGL(degree_keys_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, deg, n, keys, ids);
        BIS_OR_CHECK(sort_pairs(keys, keys_out, ids, by_key, (size_t)n));
        hipLaunchKernelGGL(rank_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, by_key, n, rank);
    } else {
        hipLaunchKernelGGL(iota_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, by_key, n);
        hipLaunchKernelGGL(iota_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, rank, n);
    }
    hipLaunchKernelGGL(fill32_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, pos, n, (int32_t)-1);
    hipLaunchKernelGGL(fill32_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, parent, n, (int32_t)INT32_MAX);
    hipLaunchKernelGGL(fill32_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, mark, n, (int32_t)0);
    int64_t count = 0, cursor = 0;
    int32_t level_id = 0;
    int components = 0;
    int *first = status + 2, *next_count = status + 4;
    while (count < n) {
        if (++components > 64) { ctx->err = "bis_mat_bfs_order: more than 64 connected components: use the host ordering"; return cleanup(BIS_ERR_UNSUPPORTED); }
        const int big = INT32_MAX;
        BIS_OR_CHECK(hipMemcpyAsync(first, &big, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(first_unseen_kernel, dim3((unsigned)std::min<int64_t>((n - cursor + 255) / 256, 1024)), dim3(256), 0, ctx->stream,
                           by_key, pos, cursor, n, first);
        hipLaunchKernelGGL(start_component_kernel, dim3(1), dim3(1), 0, ctx->stream, by_key, first, order, pos, count);
        int h_first = 0;
        BIS_OR_CHECK(hipMemcpyAsync(&h_first, first, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        BIS_OR_CHECK(hipStreamSynchronize(ctx->stream));
        cursor = (int64_t)h_first + 1;
        int64_t f0 = count, f1 = count + 1;
        count = f1;
        while (f1 > f0) {
            ++level_id;
            BIS_OR_CHECK(hipMemsetAsync(next_count, 0, sizeof(int), ctx->stream));
            hipLaunchKernelGGL(expand_kernel<RP>, dim3((unsigned)((f1 - f0 + 255) / 256)), dim3(256), 0, ctx->stream, rp, A->col, order, f0, f1,
                               pos, parent, mark, level_id, next_list, next_count);
            int m = 0;
            BIS_OR_CHECK(hipMemcpyAsync(&m, next_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            BIS_OR_CHECK(hipStreamSynchronize(ctx->stream));
            if (m == 0) break;
            const unsigned gb = (unsigned)((m + 255) / 256);
            hipLaunchKernelGGL(child_keys_kernel, dim3(gb), dim3(256), 0, ctx->stream, next_list, m, parent, rank, keys);
            BIS_OR_CHECK(sort_pairs(keys, keys_out, next_list, sorted, (size_t)m));
            hipLaunchKernelGGL(append_kernel, dim3(gb), dim3(256), 0, ctx->stream, sorted, m, count, order, pos);
            f0 = count;
            f1 = count + m;
            count = f1;
        }
    }
    if (rcm) hipLaunchKernelGGL(reverse_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, order, n, perm_dev);
    else BIS_OR_CHECK(hipMemcpyAsync(perm_dev, order, 4 * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    BIS_OR_CHECK(hipGetLastError());
    BIS_OR_CHECK(hipStreamSynchronize(ctx->stream));
    return cleanup(BIS_OK);
}
#undef BIS_OR_CHECK

} // namespace

extern "C" {

bis_status bis_mat_permute(bis_ctx *ctx, const bis_mat *A, const int32_t *perm_dev, bis_mat **B) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A && B && (A->n_rows == 0 || perm_dev), "bis_mat_permute: bad arguments");
    BIS_REQUIRE(ctx, A->n_rows == A->n_cols, "bis_mat_permute: square matrix required");
    BIS_REQUIRE(ctx, !A->view, "bis_mat_permute: owning matrix required");
    return A->rp64 ? permute_t<int64_t>(ctx, A, perm_dev, B) : permute_t<int32_t>(ctx, A, perm_dev, B);
}

bis_status bis_mat_bfs_order(bis_ctx *ctx, const bis_mat *A, int rcm, int32_t *perm_dev) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A && (A->n_rows == 0 || perm_dev), "bis_mat_bfs_order: bad arguments");
    BIS_REQUIRE(ctx, A->n_rows == A->n_cols, "bis_mat_bfs_order: square matrix required");
    return A->rp64 ? bfs_order_t<int64_t>(ctx, A, rcm != 0, perm_dev) : bfs_order_t<int32_t>(ctx, A, rcm != 0, perm_dev);
}

} // extern "C"
