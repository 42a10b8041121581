// bis_dist.hip -- 1-D row-block partition of the hot path across GPUs
// (SURVEY.md section 8e).  The reference is single-process OpenMP; this layer
// is new.  One process per GPU; rank g owns a contiguous global row range.
//
//   SpMV   : pack boundary x entries -> halo exchange (second stream, RCCL
//            ncclSend/ncclRecv pairs over xGMI, or the launcher's transport)
//            overlapped with the SpMV of the interior rows (no remote column)
//            -> SpMV of the boundary rows on [x_local | halo].
//   dot    : local two-stage reduction -> all-reduce of 1-2 doubles.
//
// Local column numbering: owned columns -> [0, n_local); the k-th distinct
// remote column (sorted by global index, hence grouped by owner) ->
// n_local + k, so the exchange receives straight into the tail of the SpMV
// input vector.
#include "bis_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>

struct bis_dist {
    int rank = 0, n_ranks = 1;
    int64_t row0 = 0, row1 = 0, n_local = 0, n_halo = 0;
    std::vector<int64_t> row_starts;
    bis_mat *A = nullptr;                 // local rows, renumbered columns (owned)
    bis_mat *lo = nullptr, *mid = nullptr, *hi = nullptr; // row views
    int64_t mid_a = 0, mid_b = 0;
    std::vector<int32_t> halo_cols;       // global indices, sorted
    std::vector<int64_t> recv_counts, send_counts;
    int64_t n_send = 0;
    int32_t *send_idx = nullptr;          // device: local indices to pack
    double *sendbuf = nullptr;            // device
    bis_comm_ops ops{};
    bool have_ops = false;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_halo = nullptr;
    // HIP-event timing of the two exchange shapes while bis_profile_enable is on (bis_dist_profile_read)
    struct EvPool {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
        size_t used = 0;
        std::pair<hipEvent_t, hipEvent_t> &next() {
            if (used == ev.size()) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); ev.emplace_back(a, b); }
            return ev[used++];
        }
        void destroy() { for (auto &p : ev) { hipEventDestroy(p.first); hipEventDestroy(p.second); } ev.clear(); used = 0; }
    } prof_exchange, prof_allreduce;
    // RCCL backend state
    void *rccl_lib = nullptr;
    ncclComm_t comm = nullptr;
    struct {
        ncclResult_t (*GetUniqueId)(ncclUniqueId *);
        ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
        ncclResult_t (*CommDestroy)(ncclComm_t);
        ncclResult_t (*CommCount)(const ncclComm_t, int *);
        ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
        ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
        ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
        ncclResult_t (*GroupStart)();
        ncclResult_t (*GroupEnd)();
        const char *(*GetErrorString)(ncclResult_t);
    } nccl{};
};

namespace {

__global__ __launch_bounds__(256) void renumber_cols_kernel(int32_t *col, int64_t nnz, int64_t row0,
                                                            int64_t row1, const int32_t *halo,
                                                            int64_t n_halo) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t n_local = row1 - row0;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnz; k += stride) {
        const int64_t c = col[k];
        if (c >= row0 && c < row1) {
            col[k] = (int32_t)(c - row0);
        } else {
            int64_t lo = 0, hi = n_halo;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (halo[mid] < c) lo = mid + 1; else hi = mid;
            }
            col[k] = (int32_t)(n_local + lo);
        }
    }
}

__global__ __launch_bounds__(256) void pack_kernel(const double *x, const int32_t *idx, int64_t n,
                                                   double *out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = x[idx[i]];
}

// ---- halo plan on the device: only what the plan needs leaves HBM ---------------------------------
// count pass: per row the number of remote entries (column outside [row0,row1)); per 256-row block
// the number of remote entries and of boundary rows (rows with at least one).
template <typename RP>
__global__ __launch_bounds__(256) void plan_count_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                         int64_t n_rows, int64_t row0, int64_t row1,
                                                         int64_t *__restrict__ blk_ent, int64_t *__restrict__ blk_row) {
    __shared__ double lds[4];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int cnt = 0;
    if (r < n_rows)
        for (int64_t k = (int64_t)row_ptr[r]; k < (int64_t)row_ptr[r + 1]; ++k) {
            const int64_t c = col[k];
            cnt += (c < row0 || c >= row1);
        }
    const double se = block_sum<256>((double)cnt, lds); // exact: counts < 2^53
    __syncthreads();
    const double sr = block_sum<256>(cnt > 0 ? 1.0 : 0.0, lds);
    if (threadIdx.x == 0) { blk_ent[blockIdx.x] = (int64_t)se; blk_row[blockIdx.x] = (int64_t)sr; }
}

// exclusive scan of the two block-sum arrays by one workgroup; totals to out2
__global__ __launch_bounds__(256) void plan_scan_kernel(int64_t *a, int64_t *b, int n_blk, int64_t *out2) {
    __shared__ int64_t sa[256], sb[256];
    int64_t run_a = 0, run_b = 0;
    for (int base = 0; base < n_blk; base += 256) {
        const int i = base + threadIdx.x;
        const int64_t va = i < n_blk ? a[i] : 0, vb = i < n_blk ? b[i] : 0;
        sa[threadIdx.x] = va;
        sb[threadIdx.x] = vb;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int64_t xa = 0, xb = 0;
            if ((int)threadIdx.x >= off) { xa = sa[threadIdx.x - off]; xb = sb[threadIdx.x - off]; }
            __syncthreads();
            sa[threadIdx.x] += xa;
            sb[threadIdx.x] += xb;
            __syncthreads();
        }
        if (i < n_blk) { a[i] = run_a + sa[threadIdx.x] - va; b[i] = run_b + sb[threadIdx.x] - vb; }
        run_a += sa[255];
        run_b += sb[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = run_a; out2[1] = run_b; }
}

// fill pass: the remote columns (row order, duplicates kept) and the boundary rows (ascending)
template <typename RP>
__global__ __launch_bounds__(256) void plan_fill_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                        int64_t n_rows, int64_t row0, int64_t row1,
                                                        const int64_t *__restrict__ blk_ent, const int64_t *__restrict__ blk_row,
                                                        int32_t *__restrict__ out_cols, int32_t *__restrict__ out_rows) {
    __shared__ int64_t se[256], sr[256];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int cnt = 0;
    if (r < n_rows)
        for (int64_t k = (int64_t)row_ptr[r]; k < (int64_t)row_ptr[r + 1]; ++k) {
            const int64_t c = col[k];
            cnt += (c < row0 || c >= row1);
        }
    se[threadIdx.x] = cnt;
    sr[threadIdx.x] = cnt > 0;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int64_t xa = 0, xb = 0;
        if ((int)threadIdx.x >= off) { xa = se[threadIdx.x - off]; xb = sr[threadIdx.x - off]; }
        __syncthreads();
        se[threadIdx.x] += xa;
        sr[threadIdx.x] += xb;
        __syncthreads();
    }
    if (r >= n_rows || cnt == 0) return;
    int64_t pe = blk_ent[blockIdx.x] + se[threadIdx.x] - cnt;
    out_rows[blk_row[blockIdx.x] + sr[threadIdx.x] - 1] = (int32_t)r;
    for (int64_t k = (int64_t)row_ptr[r]; k < (int64_t)row_ptr[r + 1]; ++k) {
        const int64_t c = col[k];
        if (c < row0 || c >= row1) out_cols[pe++] = (int32_t)c;
    }
}

// D[r] = a(r, row_offset + r): the diagonal of a row block with GLOBAL column indices (peel_diag_crs
// semantics, utilities/LU_factors.hpp:827-869: the last diagonal entry of a row wins; zero / missing
// diagonals are reported through the status word like bis_mat_split_strict)
template <typename RP>
__global__ __launch_bounds__(256) void diag_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                   const double *__restrict__ val, int64_t n_rows, int64_t row_offset,
                                                   double *D, double *D_inv, unsigned long long *status) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t g = row_offset + r;
    bool have = false;
    for (int64_t k = (int64_t)row_ptr[r]; k < (int64_t)row_ptr[r + 1]; ++k)
        if ((int64_t)col[k] == g) {
            const double v = val[k];
            have = true;
            D[r] = v;
            if (D_inv) D_inv[r] = 1.0 / v;
            if (fabs(v) < 1e-16) atomicMin(status, ((unsigned long long)(g + 1) << 1) | 0ull);
        }
    if (!have) atomicMin(status, ((unsigned long long)(g + 1) << 1) | 1ull);
}

// entries of the diagonal block (row_offset <= col < row_offset + n_rows) per row; per-block sums
template <typename RP>
__global__ __launch_bounds__(256) void block_count_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col, int64_t n_rows,
                                                          int64_t row_offset, int64_t *__restrict__ blk) {
    __shared__ double lds[4];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int cnt = 0;
    if (r < n_rows)
        for (int64_t k = (int64_t)row_ptr[r]; k < (int64_t)row_ptr[r + 1]; ++k) {
            const int64_t c = (int64_t)col[k] - row_offset;
            cnt += (c >= 0 && c < n_rows);
        }
    const double s = block_sum<256>((double)cnt, lds);
    if (threadIdx.x == 0) blk[blockIdx.x] = (int64_t)s;
}
template <typename RP, typename RPO>
__global__ __launch_bounds__(256) void block_fill_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                         const double *__restrict__ val, int64_t n_rows, int64_t row_offset,
                                                         const int64_t *__restrict__ blk, RPO *__restrict__ rpB, int32_t *__restrict__ colB,
                                                         double *__restrict__ valB) {
    __shared__ int64_t sc[256];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int cnt = 0;
    if (r < n_rows)
        for (int64_t k = (int64_t)row_ptr[r]; k < (int64_t)row_ptr[r + 1]; ++k) {
            const int64_t c = (int64_t)col[k] - row_offset;
            cnt += (c >= 0 && c < n_rows);
        }
    sc[threadIdx.x] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int64_t v = 0;
        if ((int)threadIdx.x >= off) v = sc[threadIdx.x - off];
        __syncthreads();
        sc[threadIdx.x] += v;
        __syncthreads();
    }
    if (r >= n_rows) return;
    int64_t p = blk[blockIdx.x] + sc[threadIdx.x] - cnt;
    rpB[r] = (RPO)p;
    if (r == n_rows - 1) rpB[n_rows] = (RPO)(p + cnt);
    for (int64_t k = (int64_t)row_ptr[r]; k < (int64_t)row_ptr[r + 1]; ++k) {
        const int64_t c = (int64_t)col[k] - row_offset;
        if (c >= 0 && c < n_rows) { colB[p] = (int32_t)c; valB[p++] = val[k]; }
    }
}

// Device version of bis_halo_plan: same outputs (sorted distinct remote columns, per-owner counts,
// longest interior row run) from the remote entries and boundary rows alone -- a z-slab of HPCG-512 / 8
// ships 2 x 512^2 x 9 column indices and 2 x 512^2 row indices instead of 1.8 GB of CRS structure.
bis_status device_halo_plan(bis_ctx *ctx, const bis_mat *A, int n_ranks, const int64_t *row_starts, int64_t row0,
                            int64_t row1, std::vector<int32_t> &halo, std::vector<int64_t> &recv_counts,
                            int64_t interior[2]) {
    const int64_t n = A->n_rows;
    const int n_blk = (int)((n + 255) / 256);
    halo.clear();
    recv_counts.assign(n_ranks, 0);
    interior[0] = 0; interior[1] = n;
    if (n_blk == 0) return BIS_OK;
    int64_t *blk = nullptr;
    BIS_HIP_CHECK(ctx, hipMalloc(&blk, sizeof(int64_t) * (size_t)(2 * n_blk + 2)));
    int64_t *blk_ent = blk, *blk_row = blk + n_blk, *tot = blk + 2 * n_blk;
    if (A->rp64)
        hipLaunchKernelGGL(plan_count_kernel<int64_t>, dim3(n_blk), dim3(256), 0, ctx->stream, (const int64_t *)A->row_ptr,
                           A->col, n, row0, row1, blk_ent, blk_row);
    else
        hipLaunchKernelGGL(plan_count_kernel<int32_t>, dim3(n_blk), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr,
                           A->col, n, row0, row1, blk_ent, blk_row);
    hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(256), 0, ctx->stream, blk_ent, blk_row, n_blk, tot);
    int64_t h_tot[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(h_tot, tot, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { hipFree(blk); ctx->err = std::string("halo plan: ") + hipGetErrorString(e); return BIS_ERR_HIP; }
    std::vector<int32_t> ent((size_t)h_tot[0]), rows((size_t)h_tot[1]);
    if (h_tot[0] > 0) {
        int32_t *d_ent = nullptr, *d_rows = nullptr;
        e = hipMalloc(&d_ent, sizeof(int32_t) * (size_t)h_tot[0]);
        if (e == hipSuccess) e = hipMalloc(&d_rows, sizeof(int32_t) * (size_t)h_tot[1]);
        if (e == hipSuccess) {
            if (A->rp64)
                hipLaunchKernelGGL(plan_fill_kernel<int64_t>, dim3(n_blk), dim3(256), 0, ctx->stream, (const int64_t *)A->row_ptr,
                                   A->col, n, row0, row1, blk_ent, blk_row, d_ent, d_rows);
            else
                hipLaunchKernelGGL(plan_fill_kernel<int32_t>, dim3(n_blk), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr,
                                   A->col, n, row0, row1, blk_ent, blk_row, d_ent, d_rows);
            e = hipMemcpyAsync(ent.data(), d_ent, sizeof(int32_t) * ent.size(), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(rows.data(), d_rows, sizeof(int32_t) * rows.size(), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
        hipFree(d_ent);
        hipFree(d_rows);
    }
    hipFree(blk);
    if (e != hipSuccess) { ctx->err = std::string("halo plan: ") + hipGetErrorString(e); return BIS_ERR_HIP; }
    std::sort(ent.begin(), ent.end());
    ent.erase(std::unique(ent.begin(), ent.end()), ent.end());
    int p = 0;
    for (int32_t c : ent) {
        while (p < n_ranks - 1 && c >= row_starts[p + 1]) ++p;
        if (c < row_starts[p] || c >= row_starts[p + 1]) { ctx->err = "bis_dist_create: column outside the global range"; return BIS_ERR_INVALID; }
        recv_counts[p]++;
    }
    halo.swap(ent);
    // longest run of rows without a remote column (same tie-breaking as bis_halo_plan: the first longest run)
    int64_t best_a = 0, best_b = 0, run_a = 0;
    for (int32_t r : rows) {
        if (r - run_a > best_b - best_a) { best_a = run_a; best_b = r; }
        run_a = (int64_t)r + 1;
    }
    if (n - run_a > best_b - best_a) { best_a = run_a; best_b = n; }
    interior[0] = best_a; interior[1] = best_b;
    return BIS_OK;
}

void *load_rccl() {
    // prefer a copy that is already mapped (torch ships its own librccl.so);
    // never mix two RCCL instances in one process
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return h;
}

template <typename F>
bool sym(void *lib, const char *name, F &out) {
    out = reinterpret_cast<F>(dlsym(lib, name));
    return out != nullptr;
}

bool bind_rccl(bis_dist *d) {
    if (d->rccl_lib) return true;
    void *h = load_rccl();
    if (!h) return false;
    bool ok = sym(h, "ncclGetUniqueId", d->nccl.GetUniqueId) &&
              sym(h, "ncclCommInitRank", d->nccl.CommInitRank) &&
              sym(h, "ncclCommDestroy", d->nccl.CommDestroy) && sym(h, "ncclCommCount", d->nccl.CommCount) &&
              sym(h, "ncclAllReduce", d->nccl.AllReduce) && sym(h, "ncclSend", d->nccl.Send) &&
              sym(h, "ncclRecv", d->nccl.Recv) && sym(h, "ncclGroupStart", d->nccl.GroupStart) &&
              sym(h, "ncclGroupEnd", d->nccl.GroupEnd) &&
              sym(h, "ncclGetErrorString", d->nccl.GetErrorString);
    if (ok) d->rccl_lib = h;
    return ok;
}

int rccl_allreduce(void *user, void *stream, double *buf, int count) {
    bis_dist *d = (bis_dist *)user;
    return d->nccl.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, d->comm,
                             (hipStream_t)stream) == ncclSuccess ? 0 : 1;
}

int rccl_exchange(void *user, void *stream, const double *sendbuf, const int64_t *send_counts,
                  double *recvbuf, const int64_t *recv_counts, int n_ranks) {
    bis_dist *d = (bis_dist *)user;
    hipStream_t s = (hipStream_t)stream;
    bool ok = d->nccl.GroupStart() == ncclSuccess;
    int64_t so = 0, ro = 0;
    for (int p = 0; p < n_ranks && ok; ++p) {
        if (send_counts[p] > 0)
            ok = ok && d->nccl.Send(sendbuf + so, (size_t)send_counts[p], ncclDouble, p, d->comm, s) == ncclSuccess;
        if (recv_counts[p] > 0)
            ok = ok && d->nccl.Recv(recvbuf + ro, (size_t)recv_counts[p], ncclDouble, p, d->comm, s) == ncclSuccess;
        so += send_counts[p];
        ro += recv_counts[p];
    }
    ok = (d->nccl.GroupEnd() == ncclSuccess) && ok;
    return ok ? 0 : 1;
}

} // namespace

extern "C" {

bis_status bis_halo_plan(int64_t n_local, const int64_t *row_ptr, const int32_t *col, int n_ranks,
                         int rank, const int64_t *row_starts, int64_t *n_halo, int32_t *halo_cols,
                         int64_t halo_cap, int64_t *recv_counts, int64_t *interior) {
    if (n_local < 0 || !row_ptr || n_ranks < 1 || rank < 0 || rank >= n_ranks || !row_starts || !n_halo)
        return BIS_ERR_INVALID;
    const int64_t row0 = row_starts[rank], row1 = row_starts[rank + 1];
    if (row1 - row0 != n_local) return BIS_ERR_INVALID;
    std::vector<int32_t> remote;
    int64_t best_a = 0, best_b = 0, run_a = 0;
    for (int64_t r = 0; r < n_local; ++r) {
        bool has_remote = false;
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
            const int64_t c = col[k];
            if (c < row0 || c >= row1) { remote.push_back((int32_t)c); has_remote = true; }
        }
        if (has_remote) {
            if (r - run_a > best_b - best_a) { best_a = run_a; best_b = r; }
            run_a = r + 1;
        }
    }
    if (n_local - run_a > best_b - best_a) { best_a = run_a; best_b = n_local; }
    std::sort(remote.begin(), remote.end());
    remote.erase(std::unique(remote.begin(), remote.end()), remote.end());
    *n_halo = (int64_t)remote.size();
    if (recv_counts) {
        for (int p = 0; p < n_ranks; ++p) recv_counts[p] = 0;
        int p = 0;
        for (int32_t c : remote) {
            while (p < n_ranks - 1 && c >= row_starts[p + 1]) ++p;
            if (c < row_starts[p] || c >= row_starts[p + 1]) return BIS_ERR_INVALID; // column outside the global range
            recv_counts[p]++;
        }
    }
    if (interior) { interior[0] = best_a; interior[1] = best_b; }
    if (halo_cols) {
        if (halo_cap < (int64_t)remote.size()) return BIS_ERR_INVALID;
        std::copy(remote.begin(), remote.end(), halo_cols);
    }
    return BIS_OK;
}

bis_status bis_mat_diag(bis_ctx *ctx, const bis_mat *A, int64_t row_offset, double *D, double *D_inv) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A && D && row_offset >= 0, "bis_mat_diag: bad arguments");
    if (A->n_rows == 0) return BIS_OK;
    unsigned long long *status = (unsigned long long *)(ctx->scalars_dev + 34);
    BIS_HIP_CHECK(ctx, hipMemsetAsync(status, 0xFF, 8, ctx->stream));
    const unsigned grid = (unsigned)((A->n_rows + 255) / 256);
    if (A->rp64)
        hipLaunchKernelGGL(diag_kernel<int64_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int64_t *)A->row_ptr, A->col,
                           A->val, A->n_rows, row_offset, D, D_inv, status);
    else
        hipLaunchKernelGGL(diag_kernel<int32_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col,
                           A->val, A->n_rows, row_offset, D, D_inv, status);
    unsigned long long h = 0;
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&h, status, 8, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h != ~0ull) {
        char msg[128];
        const bool missing = h & 1ull;
        snprintf(msg, sizeof msg, missing ? "No diagonal to extract at row index %lld" : "Zero detected on diagonal at row index %lld",
                 (long long)(h >> 1) - 1);
        ctx->err = msg;
        return missing ? BIS_ERR_NO_DIAG : BIS_ERR_ZERO_DIAG;
    }
    return BIS_OK;
}

} // extern "C"

// The diagonal block of a row block with GLOBAL column indices: the entries with row_offset <= col < row_offset +
// n_rows, columns renumbered to [0, n_rows), order inside a row kept.
template <typename RP>
static bis_status diag_block_t(bis_ctx *ctx, const bis_mat *A, int64_t row_offset, bis_mat **out) {
    const int64_t n = A->n_rows;
    const int n_blk = (int)((n + 255) / 256);
    int64_t *blk = nullptr;
    BIS_HIP_CHECK(ctx, hipMalloc(&blk, sizeof(int64_t) * (size_t)(2 * n_blk + 4)));
    int64_t *blk_ent = blk, *dummy = blk + n_blk, *tot = blk + 2 * n_blk;
    BIS_HIP_CHECK(ctx, hipMemsetAsync(blk, 0, sizeof(int64_t) * (size_t)(2 * n_blk + 4), ctx->stream));
    if (n_blk) {
        hipLaunchKernelGGL((block_count_kernel<RP>), dim3(n_blk), dim3(256), 0, ctx->stream, (const RP *)A->row_ptr, A->col, n, row_offset, blk_ent);
        hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(256), 0, ctx->stream, blk_ent, dummy, n_blk, tot);
    }
    int64_t h_tot[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(h_tot, tot, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { hipFree(blk); ctx->err = "bis_mat_diag_block: count pass failed"; return BIS_ERR_HIP; }
    bis_mat *B = nullptr;
    bis_status st = bis_mat_alloc(ctx, n, n, h_tot[0], bis_want_rp64(h_tot[0]), &B);
    if (st != BIS_OK) { hipFree(blk); return st; }
    if (n_blk) {
        if (B->rp64)
            hipLaunchKernelGGL((block_fill_kernel<RP, int64_t>), dim3(n_blk), dim3(256), 0, ctx->stream, (const RP *)A->row_ptr, A->col, A->val, n,
                               row_offset, blk_ent, (int64_t *)B->row_ptr, B->col, B->val);
        else
            hipLaunchKernelGGL((block_fill_kernel<RP, int32_t>), dim3(n_blk), dim3(256), 0, ctx->stream, (const RP *)A->row_ptr, A->col, A->val, n,
                               row_offset, blk_ent, (int32_t *)B->row_ptr, B->col, B->val);
    } else {
        hipMemsetAsync(B->row_ptr, 0, B->rp64 ? 8 : 4, ctx->stream);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(blk);
    if (e != hipSuccess) { bis_mat_destroy(ctx, B); ctx->err = "bis_mat_diag_block: fill pass failed"; return BIS_ERR_HIP; }
    st = bis_mat_finalize(ctx, B);
    if (st != BIS_OK) { bis_mat_destroy(ctx, B); return st; }
    *out = B;
    return BIS_OK;
}

extern "C" {

bis_status bis_mat_diag_block(bis_ctx *ctx, const bis_mat *A_local, int64_t row_offset, bis_mat **block) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A_local && block && row_offset >= 0, "bis_mat_diag_block: bad arguments");
    return A_local->rp64 ? diag_block_t<int64_t>(ctx, A_local, row_offset, block) : diag_block_t<int32_t>(ctx, A_local, row_offset, block);
}

bis_status bis_dist_create(bis_ctx *ctx, bis_mat *A, int rank, int n_ranks, const int64_t *row_starts,
                           bis_dist **out) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A && out && n_ranks >= 1 && rank >= 0 && rank < n_ranks && row_starts,
                "bis_dist_create: bad arguments");
    BIS_REQUIRE(ctx, !A->view, "bis_dist_create: A_local must own its arrays");
    const int64_t row0 = row_starts[rank], row1 = row_starts[rank + 1];
    BIS_REQUIRE(ctx, row1 - row0 == A->n_rows, "bis_dist_create: row range does not match A_local");
    BIS_REQUIRE(ctx, row_starts[n_ranks] == A->n_cols, "bis_dist_create: A_local.n_cols must be the global size");
    bis_dist *d = new bis_dist;
    d->rank = rank;
    d->n_ranks = n_ranks;
    d->row0 = row0;
    d->row1 = row1;
    d->n_local = A->n_rows;
    d->row_starts.assign(row_starts, row_starts + n_ranks + 1);
    d->recv_counts.assign(n_ranks, 0);
    d->send_counts.assign(n_ranks, 0);
    // plan: on the device (only the remote entries and the boundary rows are downloaded); the host planner
    // (bis_halo_plan on the downloaded structure) stays selectable for cross-checks (dist_host_plan=1)
    int64_t interior[2] = {0, 0};
    bis_status st;
    if (bis_opts().dist_host_plan > 0) {
        std::vector<int64_t> rp(A->n_rows + 1);
        std::vector<int32_t> col((size_t)std::max<int64_t>(A->nnz, 1));
        st = bis_mat_download(ctx, A, rp.data(), col.data(), nullptr);
        if (st == BIS_OK) {
            st = bis_halo_plan(A->n_rows, rp.data(), col.data(), n_ranks, rank, row_starts, &d->n_halo,
                               nullptr, 0, d->recv_counts.data(), interior);
            if (st == BIS_OK) {
                d->halo_cols.resize((size_t)d->n_halo);
                st = bis_halo_plan(A->n_rows, rp.data(), col.data(), n_ranks, rank, row_starts, &d->n_halo,
                                   d->halo_cols.data(), d->n_halo, d->recv_counts.data(), interior);
            }
            if (st != BIS_OK) ctx->err = "bis_dist_create: halo planning failed (column outside the global range?)";
        }
    } else {
        st = device_halo_plan(ctx, A, n_ranks, row_starts, row0, row1, d->halo_cols, d->recv_counts, interior);
        d->n_halo = (int64_t)d->halo_cols.size();
    }
    if (st != BIS_OK) { delete d; return st; }
    // renumber the columns on the device
    int32_t *halo_dev = nullptr;
    hipError_t e = hipMalloc(&halo_dev, sizeof(int32_t) * (size_t)std::max<int64_t>(d->n_halo, 1));
    if (e == hipSuccess && d->n_halo)
        e = hipMemcpyAsync(halo_dev, d->halo_cols.data(), sizeof(int32_t) * (size_t)d->n_halo,
                           hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && A->nnz) {
        const int grid = (int)std::min<int64_t>((A->nnz + 255) / 256, 8192);
        hipLaunchKernelGGL(renumber_cols_kernel, dim3(grid), dim3(256), 0, ctx->stream, A->col, A->nnz,
                           row0, row1, halo_dev, d->n_halo);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(halo_dev);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_packed, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_halo, hipEventDisableTiming);
    if (e != hipSuccess) {
        ctx->err = std::string("bis_dist_create: ") + hipGetErrorString(e);
        delete d;
        return BIS_ERR_HIP;
    }
    A->n_cols = d->n_local + d->n_halo;
    st = bis_mat_finalize(ctx, A); // the column indices changed: rebuild the packed stream / window structures
    if (st != BIS_OK) { hipEventDestroy(d->ev_packed); hipEventDestroy(d->ev_halo); hipStreamDestroy(d->comm_stream); delete d; return st; }
    if (bis_opts().tune_placement > 0) { // optional setup step, before the views share A's arrays
        st = bis_mat_tune_placement(ctx, A, bis_opts().tune_placement, nullptr, nullptr);
        if (st != BIS_OK) { hipEventDestroy(d->ev_packed); hipEventDestroy(d->ev_halo); hipStreamDestroy(d->comm_stream); delete d; return st; }
    }
    d->A = A;
    d->mid_a = interior[0];
    d->mid_b = interior[1];
    st = bis_mat_row_view(ctx, A, 0, d->mid_a, &d->lo);
    if (st == BIS_OK) st = bis_mat_row_view(ctx, A, d->mid_a, d->mid_b, &d->mid);
    if (st == BIS_OK) st = bis_mat_row_view(ctx, A, d->mid_b, A->n_rows, &d->hi);
    if (st != BIS_OK) { bis_dist_destroy(ctx, d); return st; }
    *out = d;
    return BIS_OK;
}

bis_status bis_dist_destroy(bis_ctx *ctx, bis_dist *d) {
    BIS_CTX_OK(ctx);
    if (!d) return BIS_OK;
    hipStreamSynchronize(ctx->stream);
    if (d->comm_stream) hipStreamSynchronize(d->comm_stream);
    if (d->comm && d->nccl.CommDestroy) d->nccl.CommDestroy(d->comm);
    bis_mat_destroy(ctx, d->lo);
    bis_mat_destroy(ctx, d->mid);
    bis_mat_destroy(ctx, d->hi);
    bis_mat_destroy(ctx, d->A);
    hipFree(d->send_idx);
    hipFree(d->sendbuf);
    if (d->ev_packed) hipEventDestroy(d->ev_packed);
    if (d->ev_halo) hipEventDestroy(d->ev_halo);
    if (d->comm_stream) hipStreamDestroy(d->comm_stream);
    d->prof_exchange.destroy();
    d->prof_allreduce.destroy();
    delete d;
    return BIS_OK;
}

bis_status bis_dist_stats(const bis_dist *d, int64_t *n_halo, int64_t *n_send, int64_t *interior_rows,
                          int *n_neighbours, int *rccl_ranks) {
    if (d && rccl_ranks) {
        int k = 0;
        if (d->comm && d->nccl.CommCount && d->nccl.CommCount(d->comm, &k) != ncclSuccess) k = -1;
        *rccl_ranks = k; // 0: the transport is not the native RCCL one
    }
    if (!d) return BIS_ERR_INVALID;
    if (n_halo) *n_halo = d->n_halo;
    if (n_send) *n_send = d->n_send;
    if (interior_rows) *interior_rows = d->mid_b - d->mid_a;
    if (n_neighbours) {
        int k = 0;
        for (int p = 0; p < d->n_ranks; ++p) k += (d->recv_counts[p] > 0 || d->send_counts[p] > 0);
        *n_neighbours = k;
    }
    return BIS_OK;
}

bis_status bis_dist_spmv_stream_info(bis_ctx *ctx, const bis_dist *d, int *col_bytes, int *val_bytes, int *n_dict, int *form) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, d && d->mid, "bis_dist_spmv_stream_info: null handle");
    return bis_mat_spmv_stream_info(ctx, d->mid, col_bytes, val_bytes, n_dict, form);
}

bis_status bis_dist_spmv_streamed_bytes(bis_ctx *ctx, const bis_dist *d, int64_t *bytes) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, d && d->mid && bytes, "bis_dist_spmv_streamed_bytes: bad arguments");
    int64_t tot = 8 * d->A->n_cols; // x = [owned | halo], once for the three launches
    for (const bis_mat *v : {d->lo, d->mid, d->hi}) {
        int64_t b = 0;
        if (v->n_rows == 0) continue;
        if (bis_status st = bis_mat_spmv_streamed_bytes(ctx, v, &b)) return st;
        tot += b - 8 * v->n_cols;
    }
    *bytes = tot;
    return BIS_OK;
}

bis_status bis_dist_profile_read(bis_ctx *ctx, bis_dist *d, int64_t *n_exchange, double *exchange_ms,
                                 int64_t *n_allreduce, double *allreduce_ms) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, d, "bis_dist_profile_read: null handle");
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(d->comm_stream));
    auto total = [](bis_dist::EvPool &p, int64_t *n, double *ms) {
        double t = 0.0;
        for (size_t i = 0; i < p.used; ++i) { float f = 0.f; if (hipEventElapsedTime(&f, p.ev[i].first, p.ev[i].second) == hipSuccess) t += f; }
        if (n) *n = (int64_t)p.used;
        if (ms) *ms = t;
        p.used = 0;
    };
    total(d->prof_exchange, n_exchange, exchange_ms);
    total(d->prof_allreduce, n_allreduce, allreduce_ms);
    return BIS_OK;
}

bis_status bis_dist_vec_len(const bis_dist *d, int64_t *n_local, int64_t *n_ext) {
    if (!d) return BIS_ERR_INVALID;
    if (n_local) *n_local = d->n_local;
    if (n_ext) *n_ext = d->n_local + d->n_halo;
    return BIS_OK;
}

bis_status bis_dist_halo_info(const bis_dist *d, int64_t *n_halo, int32_t *halo_cols, int64_t halo_cap,
                              int64_t *recv_counts) {
    if (!d) return BIS_ERR_INVALID;
    if (n_halo) *n_halo = d->n_halo;
    if (halo_cols) {
        if (halo_cap < d->n_halo) return BIS_ERR_INVALID;
        std::copy(d->halo_cols.begin(), d->halo_cols.end(), halo_cols);
    }
    if (recv_counts) std::copy(d->recv_counts.begin(), d->recv_counts.end(), recv_counts);
    return BIS_OK;
}

bis_status bis_dist_set_send_lists(bis_ctx *ctx, bis_dist *d, const int64_t *send_counts,
                                   const int32_t *send_cols_global) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, d && send_counts, "bis_dist_set_send_lists: bad arguments");
    int64_t total = 0;
    for (int p = 0; p < d->n_ranks; ++p) {
        BIS_REQUIRE(ctx, send_counts[p] >= 0, "bis_dist_set_send_lists: negative count");
        d->send_counts[p] = send_counts[p];
        total += send_counts[p];
    }
    BIS_REQUIRE(ctx, total == 0 || send_cols_global, "bis_dist_set_send_lists: null list");
    std::vector<int32_t> idx((size_t)std::max<int64_t>(total, 1));
    for (int64_t i = 0; i < total; ++i) {
        const int64_t c = send_cols_global[i];
        BIS_REQUIRE(ctx, c >= d->row0 && c < d->row1,
                    "bis_dist_set_send_lists: requested column is not owned by this rank");
        idx[i] = (int32_t)(c - d->row0);
    }
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    hipFree(d->send_idx);
    hipFree(d->sendbuf);
    d->send_idx = nullptr;
    d->sendbuf = nullptr;
    BIS_HIP_CHECK(ctx, hipMalloc(&d->send_idx, sizeof(int32_t) * idx.size()));
    BIS_HIP_CHECK(ctx, hipMalloc(&d->sendbuf, sizeof(double) * idx.size()));
    BIS_HIP_CHECK(ctx, hipMemcpy(d->send_idx, idx.data(), sizeof(int32_t) * idx.size(), hipMemcpyHostToDevice));
    d->n_send = total;
    return BIS_OK;
}

bis_status bis_dist_set_comm(bis_ctx *ctx, bis_dist *d, const bis_comm_ops *ops) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, d && ops && ops->allreduce_sum && ops->exchange, "bis_dist_set_comm: bad arguments");
    d->ops = *ops;
    d->have_ops = true;
    return BIS_OK;
}

bis_status bis_rccl_unique_id(bis_ctx *ctx, void *out128) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, out128, "bis_rccl_unique_id: null output");
    bis_dist tmp;
    if (!bind_rccl(&tmp)) { ctx->err = "RCCL library not found (librccl.so.1)"; return BIS_ERR_COMM; }
    ncclUniqueId id;
    if (tmp.nccl.GetUniqueId(&id) != ncclSuccess) { ctx->err = "ncclGetUniqueId failed"; return BIS_ERR_COMM; }
    static_assert(sizeof(id) == 128, "ncclUniqueId size");
    memcpy(out128, &id, 128);
    return BIS_OK;
}

bis_status bis_dist_use_rccl(bis_ctx *ctx, bis_dist *d, const void *unique_id128) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, d && unique_id128, "bis_dist_use_rccl: bad arguments");
    if (!bind_rccl(d)) { ctx->err = "RCCL library not found (librccl.so.1)"; return BIS_ERR_COMM; }
    ncclUniqueId id;
    memcpy(&id, unique_id128, 128);
    BIS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const ncclResult_t r = d->nccl.CommInitRank(&d->comm, d->n_ranks, id, d->rank);
    if (r != ncclSuccess) {
        ctx->err = std::string("ncclCommInitRank: ") + d->nccl.GetErrorString(r);
        return BIS_ERR_COMM;
    }
    d->ops.user = d;
    d->ops.allreduce_sum = rccl_allreduce;
    d->ops.exchange = rccl_exchange;
    d->have_ops = true;
    return BIS_OK;
}

} // extern "C"

// internal (also used by the distributed CG): SpMV with the halo exchange
// overlapped; w/partials as in bis_spmv_launch (partials of the three row
// ranges are laid out back to back, *n_partials = total).
bis_status bis_dist_spmv_launch(bis_ctx *ctx, bis_dist *d, double *x_ext, double *y, const double *w,
                                int *n_partials) {
    BIS_REQUIRE(ctx, d->n_ranks == 1 || d->have_ops, "bis_dist_spmv: no transport set (bis_dist_set_comm / bis_dist_use_rccl)");
    BIS_REQUIRE(ctx, d->n_ranks == 1 || d->n_send == 0 || d->send_idx, "bis_dist_spmv: send lists not set");
    int np = 0, tot = 0;
    const bool exch = d->n_ranks > 1 && (d->n_send > 0 || d->n_halo > 0);
    if (exch) {
        if (d->n_send > 0) {
            const int grid = (int)std::min<int64_t>((d->n_send + 255) / 256, 2048);
            hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(256), 0, ctx->stream, x_ext, d->send_idx,
                               d->n_send, d->sendbuf);
        }
        BIS_HIP_CHECK(ctx, hipEventRecord(d->ev_packed, ctx->stream));
        BIS_HIP_CHECK(ctx, hipStreamWaitEvent(d->comm_stream, d->ev_packed, 0));
        std::pair<hipEvent_t, hipEvent_t> *pe = ctx->profile ? &d->prof_exchange.next() : nullptr;
        if (pe) hipEventRecord(pe->first, d->comm_stream);
        if (d->ops.exchange(d->ops.user, (void *)d->comm_stream, d->sendbuf, d->send_counts.data(),
                            x_ext + d->n_local, d->recv_counts.data(), d->n_ranks) != 0) {
            ctx->err = "halo exchange failed";
            return BIS_ERR_COMM;
        }
        if (pe) hipEventRecord(pe->second, d->comm_stream);
        BIS_HIP_CHECK(ctx, hipEventRecord(d->ev_halo, d->comm_stream));
    }
    // interior rows: no remote column, runs under the exchange
    bis_status st = bis_spmv_launch(ctx, d->mid, x_ext, y + d->mid_a, w ? w + d->mid_a : nullptr, &np, 0);
    if (st != BIS_OK) return st;
    tot += np;
    if (exch) BIS_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, d->ev_halo, 0));
    st = bis_spmv_launch(ctx, d->lo, x_ext, y, w, &np, (size_t)tot);
    if (st != BIS_OK) return st;
    tot += np;
    st = bis_spmv_launch(ctx, d->hi, x_ext, y + d->mid_b, w ? w + d->mid_b : nullptr, &np, (size_t)tot);
    if (st != BIS_OK) return st;
    tot += np;
    if (n_partials) *n_partials = tot;
    return BIS_OK;
}

int bis_dist_total_blocks(const bis_dist *d) { return d->lo->n_blocks_f + d->mid->n_blocks_f + d->hi->n_blocks_f; }
// Partials the fused dot of one distributed SpMV may write: the three row ranges lay theirs back to back, each with whatever its
// stream format emits -- one per wave of a row block (<= 16 per block), one per 256-row block x 4 (lane-per-row dictionary form) or
// one per 64-row slice rounded up to blocks of 512 rows (sliced-ELL form).  An upper bound over the formats, per range.
size_t bis_dist_partials_need(const bis_dist *d) {
    size_t tot = 0;
    for (const bis_mat *v : {d->lo, d->mid, d->hi}) {
        const size_t rows = (size_t)std::max<int64_t>(v->n_rows, 0);
        const size_t sell = ((rows + 511) / 512) * 8 + 8, rowmajor = ((rows + 255) / 256) * 4 + 8, rowblock = (size_t)v->n_blocks_f * 16;
        tot += std::max(sell, std::max(rowmajor, rowblock));
    }
    return tot;
}
int64_t bis_dist_n_local(const bis_dist *d) { return d->n_local; }
int64_t bis_dist_n_ext(const bis_dist *d) { return d->n_local + d->n_halo; }
const bis_mat *bis_dist_matrix(const bis_dist *d) { return d->A; }
bis_status bis_dist_allreduce(bis_ctx *ctx, bis_dist *d, double *buf_dev, int count) {
    if (d->n_ranks == 1) return BIS_OK;
    BIS_REQUIRE(ctx, d->have_ops, "bis_dist: no transport set");
    std::pair<hipEvent_t, hipEvent_t> *pe = ctx->profile ? &d->prof_allreduce.next() : nullptr;
    if (pe) hipEventRecord(pe->first, ctx->stream);
    if (d->ops.allreduce_sum(d->ops.user, (void *)ctx->stream, buf_dev, count) != 0) {
        ctx->err = "all-reduce failed";
        return BIS_ERR_COMM;
    }
    if (pe) hipEventRecord(pe->second, ctx->stream);
    return BIS_OK;
}

extern "C" {

bis_status bis_dist_spmv(bis_ctx *ctx, bis_dist *d, double *x_ext, double *y_local) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, d && x_ext && y_local, "bis_dist_spmv: bad arguments");
    return bis_dist_spmv_launch(ctx, d, x_ext, y_local, nullptr, nullptr);
}

bis_status bis_dist_dot(bis_ctx *ctx, bis_dist *d, const double *a, const double *b,
                        double *result_dev, double *result_host) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, d && result_dev, "bis_dist_dot: bad arguments");
    bis_status st = bis_dot_dev(ctx, a, b, d->n_local, result_dev);
    if (st == BIS_OK) st = bis_dist_allreduce(ctx, d, result_dev, 1);
    if (st != BIS_OK) return st;
    if (result_host) {
        BIS_HIP_CHECK(ctx, hipMemcpyAsync(ctx->scalars_host, result_dev, sizeof(double),
                                          hipMemcpyDeviceToHost, ctx->stream));
        BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        *result_host = ctx->scalars_host[0];
    }
    return BIS_OK;
}

} // extern "C"
