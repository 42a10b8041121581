// bis_matrix.hip -- device-resident MatrixCRS (reference sparse_matrix.hpp:
// 59-179), synthetic-input generators that build the CRS directly in HBM
// (SURVEY.md section 8d), SpMV row-block metadata, and the device version of
// the strict-triangle split + diagonal extraction (utilities/LU_factors.hpp:
// 122-309, :827-869).
//
// HBM layout: row_ptr int32 (int64 when nnz >= 2^31), col int32, val fp64 --
// the reference's layout (sparse_matrix.hpp:60-66), column order within a row
// exactly as given.  col/val carry 8 elements of zero padding so that the
// SpMV kernel's 16-byte vector loads may start 4-aligned before a block's
// first non-zero and run past its last one.
#include "bis_internal.hpp"

#include <algorithm>
#include <utility>
#include <vector>

#include <cstdlib>

namespace {

constexpr int kPad = 8;

template <typename RP>
__global__ void max_row_kernel(const RP *row_ptr, int64_t n_rows, int *out_max) {
    int m = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows;
         r += (int64_t)gridDim.x * blockDim.x)
        m = max(m, (int)(row_ptr[r + 1] - row_ptr[r]));
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out_max, m);
}

// Row blocks balanced on weight(r) = (row_ptr[r] - row_ptr[0]) + r, i.e. non-zeros
// plus rows before row r: blk_row[k] = first row with weight >= k*chunk, so a
// block holds at most ~chunk non-zeros AND at most chunk rows (long runs of
// empty rows -- strict triangles of colour-sorted matrices -- still spread over
// many workgroups).  blk_row[n_blocks] = n_rows.
template <typename RP>
__global__ void row_blocks_kernel(const RP *row_ptr, int64_t n_rows, int n_blocks,
                                  int64_t chunk, int32_t *blk_row, int64_t *blk_nnz) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_blocks) return;
    if (k == n_blocks) { blk_row[k] = (int32_t)n_rows; blk_nnz[k] = (int64_t)row_ptr[n_rows]; return; }
    const int64_t base = (int64_t)row_ptr[0]; // views: row_ptr[0] != 0
    const int64_t target = (int64_t)k * chunk;
    int64_t lo = 0, hi = n_rows; // answer in [0, n_rows]
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)row_ptr[mid] - base + mid >= target) hi = mid; else lo = mid + 1;
    }
    blk_row[k] = (int32_t)lo;
    blk_nnz[k] = (int64_t)row_ptr[lo];
}

// structure check of an uploaded CRS: status |= 1 row_ptr not monotone / wrong ends, |= 2 column out of range
template <typename RP>
__global__ __launch_bounds__(256) void validate_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col, int64_t n_rows,
                                                       int64_t n_cols, int64_t nnz, int *status) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t a = (int64_t)row_ptr[r], b = (int64_t)row_ptr[r + 1];
    int bad = 0;
    if (a > b || a < 0 || b > nnz || (r == 0 && a != 0) || (r == n_rows - 1 && b != nnz)) bad |= 1;
    else
        for (int64_t k = a; k < b; ++k) {
            const int64_t c = col[k];
            if (c < 0 || c >= n_cols) bad |= 2;
        }
    if (bad) atomicOr(status, bad);
}

// ---- generators ---------------------------------------------------------------
__device__ __forceinline__ int64_t hpcg_c(int64_t j, int64_t n) { return 1 + (j > 0) + (j < n - 1); }
__device__ __forceinline__ int64_t hpcg_f(int64_t k, int64_t n) {
    const int64_t a = k - 1 > 0 ? k - 1 : 0;
    const int64_t b = k < n - 1 ? k : n - 1;
    return k + a + b;
}
__host__ __device__ inline int64_t hpcg_f_h(int64_t k, int64_t n) {
    const int64_t a = k - 1 > 0 ? k - 1 : 0;
    const int64_t b = k < n - 1 ? k : n - 1;
    return k + a + b;
}
__device__ __forceinline__ int64_t hpcg_row_ptr(int64_t row, int64_t nx, int64_t ny, int64_t nz) {
    const int64_t Sx = hpcg_f(nx, nx), Sy = hpcg_f(ny, ny);
    if (row >= nx * ny * nz) return hpcg_f(nz, nz) * Sy * Sx;
    const int64_t x = row % nx, y = (row / nx) % ny, z = row / (nx * ny);
    return hpcg_f(z, nz) * Sy * Sx + hpcg_c(z, nz) * (hpcg_f(y, ny) * Sx + hpcg_c(y, ny) * hpcg_f(x, nx));
}
static int64_t hpcg_row_ptr_host(int64_t row, int64_t nx, int64_t ny, int64_t nz) {
    auto c = [](int64_t j, int64_t n) { return (int64_t)(1 + (j > 0) + (j < n - 1)); };
    const int64_t Sx = hpcg_f_h(nx, nx), Sy = hpcg_f_h(ny, ny);
    if (row >= nx * ny * nz) return hpcg_f_h(nz, nz) * Sy * Sx;
    const int64_t x = row % nx, y = (row / nx) % ny, z = row / (nx * ny);
    return hpcg_f_h(z, nz) * Sy * Sx + c(z, nz) * (hpcg_f_h(y, ny) * Sx + c(y, ny) * hpcg_f_h(x, nx));
}

template <typename RP>
__global__ void gen_hpcg_kernel(int64_t nx, int64_t ny, int64_t nz, int64_t row0, int64_t row1,
                                int64_t base, RP *row_ptr, int32_t *col, double *val) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = row0 + i;
    if (row > row1) return;
    int64_t p = hpcg_row_ptr(row, nx, ny, nz) - base;
    row_ptr[i] = (RP)p;
    if (row == row1) return;
    const int64_t x = row % nx, y = (row / nx) % ny, z = row / (nx * ny);
    for (int dz = -1; dz <= 1; ++dz) {
        if (z + dz < 0 || z + dz >= nz) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            if (y + dy < 0 || y + dy >= ny) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                if (x + dx < 0 || x + dx >= nx) continue;
                col[p] = (int32_t)(row + dx + nx * (dy + ny * (int64_t)dz));
                val[p] = (dx == 0 && dy == 0 && dz == 0) ? 26.0 : -1.0;
                ++p;
            }
        }
    }
}

__device__ __forceinline__ double anderson_u01(uint64_t seed, uint64_t i) {
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + (i + 1) * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

template <typename RP>
__global__ void gen_anderson_kernel(int64_t L, double t, double W, double shift, uint64_t seed,
                                    int64_t row0, int64_t row1, RP *row_ptr, int32_t *col,
                                    double *val) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = row0 + i;
    if (row > row1) return;
    row_ptr[i] = (RP)(i * 7);
    if (row == row1) return;
    const int64_t x = row % L, y = (row / L) % L, z = row / (L * L);
    const int64_t xm = (x + L - 1) % L, xp = (x + 1) % L;
    const int64_t ym = (y + L - 1) % L, yp = (y + 1) % L;
    const int64_t zm = (z + L - 1) % L, zp = (z + 1) % L;
    int64_t c[7];
    double v[7];
    c[0] = x + L * (y + L * zm);
    c[1] = x + L * (ym + L * z);
    c[2] = xm + L * (y + L * z);
    c[3] = row;
    c[4] = xp + L * (y + L * z);
    c[5] = x + L * (yp + L * z);
    c[6] = x + L * (y + L * zp);
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = -t;
    v[3] = fma(W, anderson_u01(seed, (uint64_t)row) - 0.5, shift);
#pragma unroll
    for (int a = 1; a < 7; ++a) { // insertion sort by column
        const int64_t ck = c[a];
        const double vk = v[a];
        int b = a - 1;
        while (b >= 0 && c[b] > ck) { c[b + 1] = c[b]; v[b + 1] = v[b]; --b; }
        c[b + 1] = ck;
        v[b + 1] = vk;
    }
    const int64_t p = i * 7;
    for (int k = 0; k < 7; ++k) { col[p + k] = (int32_t)c[k]; val[p + k] = v[k]; }
}

// ---- FEM-like unstructured generator (stand-in for SuiteSparse Flan_1565) ----
// nx*ny*nz nodes, 3 unknowns per node; node a couples (full 3x3 blocks) to itself and
// to each in-grid 27-point neighbour b that survives a symmetric coin flip; symmetric
// negative off-diagonals, a_rr = 1 + sum |a_rc| accumulated left to right (SPD).
// Same formulas, same summation order as oracle/bis_oracle.c orc_gen_fem.
constexpr uint64_t kFemK1 = 0x5851F42D4C957F2Dull, kFemK2 = 0x14057B7EF767814Full;

// one row: returns its length; writes columns/values when col != nullptr
__device__ inline int fem_row(int64_t nx, int64_t ny, int64_t nz, int keep, uint64_t seed, int64_t row,
                              int32_t *col, double *val) {
    const int64_t n_nodes = nx * ny * nz, n_rows = 3 * n_nodes;
    const int64_t a = row / 3;
    const int64_t i = a % nx, j = (a / nx) % ny, k = a / (nx * ny);
    int len = 0, dpos = -1;
    double acc = 0.0;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int64_t ii = i + dx, jj = j + dy, kk = k + dz;
                if (ii < 0 || ii >= nx || jj < 0 || jj >= ny || kk < 0 || kk >= nz) continue;
                const int64_t b = ii + nx * (jj + ny * kk);
                if (b != a) {
                    const int64_t lo = a < b ? a : b, hi = a < b ? b : a;
                    if (!(anderson_u01(seed ^ kFemK1, (uint64_t)(lo * n_nodes + hi)) * 100.0 < (double)keep)) continue;
                }
                for (int d = 0; d < 3; ++d) {
                    const int64_t c = 3 * b + d;
                    if (col) {
                        col[len] = (int32_t)c;
                        if (c == row) { dpos = len; val[len] = 0.0; }
                        else {
                            const int64_t lo = row < c ? row : c, hi = row < c ? c : row;
                            const double v = -(0.05 + 0.95 * anderson_u01(seed ^ kFemK2, (uint64_t)(lo * n_rows + hi)));
                            val[len] = v;
                            acc += fabs(v);
                        }
                    }
                    ++len;
                }
            }
    if (col) val[dpos] = 1.0 + acc;
    return len;
}

__global__ __launch_bounds__(256) void fem_count_kernel(int64_t nx, int64_t ny, int64_t nz, int keep,
                                                        uint64_t seed, int64_t row0, int64_t n_local,
                                                        int64_t *blk) {
    __shared__ double lds[4];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int len = r < n_local ? fem_row(nx, ny, nz, keep, seed, row0 + r, nullptr, nullptr) : 0;
    const double tot = block_sum<256>((double)len, lds);
    if (threadIdx.x == 0) blk[blockIdx.x] = (int64_t)tot;
}

template <typename RP>
__global__ __launch_bounds__(256) void fem_fill_kernel(int64_t nx, int64_t ny, int64_t nz, int keep,
                                                       uint64_t seed, int64_t row0, int64_t n_local,
                                                       const int64_t *blk, RP *row_ptr, int32_t *col,
                                                       double *val) {
    __shared__ int64_t sc[256];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int len = r < n_local ? fem_row(nx, ny, nz, keep, seed, row0 + r, nullptr, nullptr) : 0;
    sc[threadIdx.x] = len;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int64_t a = 0;
        if ((int)threadIdx.x >= off) a = sc[threadIdx.x - off];
        __syncthreads();
        sc[threadIdx.x] += a;
        __syncthreads();
    }
    if (r >= n_local) return;
    const int64_t p = blk[blockIdx.x] + sc[threadIdx.x] - len;
    row_ptr[r] = (RP)p;
    if (r == n_local - 1) row_ptr[n_local] = (RP)(p + len);
    fem_row(nx, ny, nz, keep, seed, row0 + r, col + p, val + p);
}

// ---- strict split ---------------------------------------------------------------
constexpr int kSplitT = 256;

// per-row counts of strict-lower / strict-upper entries; per-block sums.
// row0 = global index of local row 0 (columns are global).
template <typename RP>
__global__ __launch_bounds__(kSplitT) void split_count_kernel(const RP *row_ptr,
                                                              const int32_t *col, int64_t n_rows,
                                                              int64_t row0, int64_t *blk_l,
                                                              int64_t *blk_u) {
    __shared__ double lds[kSplitT / 64];
    const int64_t r = (int64_t)blockIdx.x * kSplitT + threadIdx.x;
    int cl = 0, cu = 0;
    if (r < n_rows) {
        const int64_t g = row0 + r;
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
            const int64_t c = col[k];
            cl += c < g;
            cu += c > g;
        }
    }
    // counts fit a double exactly (< 2^53)
    const double sl = block_sum<kSplitT>((double)cl, lds);
    __syncthreads();
    const double su = block_sum<kSplitT>((double)cu, lds);
    if (threadIdx.x == 0) { blk_l[blockIdx.x] = (int64_t)sl; blk_u[blockIdx.x] = (int64_t)su; }
}

// exclusive scan of the block sums, single workgroup; totals to out2.
__global__ __launch_bounds__(256) void split_scan_kernel(int64_t *blk_l, int64_t *blk_u, int n_blk,
                                                         int64_t *out2) {
    __shared__ int64_t sl[256], su[256];
    int64_t run_l = 0, run_u = 0;
    for (int base = 0; base < n_blk; base += 256) {
        const int i = base + threadIdx.x;
        const int64_t vl = i < n_blk ? blk_l[i] : 0, vu = i < n_blk ? blk_u[i] : 0;
        sl[threadIdx.x] = vl;
        su[threadIdx.x] = vu;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int64_t al = 0, au = 0;
            if ((int)threadIdx.x >= off) { al = sl[threadIdx.x - off]; au = su[threadIdx.x - off]; }
            __syncthreads();
            sl[threadIdx.x] += al;
            su[threadIdx.x] += au;
            __syncthreads();
        }
        if (i < n_blk) { blk_l[i] = run_l + sl[threadIdx.x] - vl; blk_u[i] = run_u + su[threadIdx.x] - vu; }
        run_l += sl[255];
        run_u += su[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = run_l; out2[1] = run_u; }
}

// status word: 0 ok, else (1+row)<<1 | kind  (kind 0 = zero diag, 1 = no diag)
template <typename RP, typename RPO>
__global__ __launch_bounds__(kSplitT) void split_fill_kernel(
    const RP *row_ptr, const int32_t *col, const double *val, int64_t n_rows, int64_t row0,
    const int64_t *blk_l, const int64_t *blk_u, RPO *rpL, int32_t *colL, double *valL, RPO *rpU,
    int32_t *colU, double *valU, double *D, double *D_inv, unsigned long long *status) {
    __shared__ int64_t sl[kSplitT], su[kSplitT];
    const int64_t r = (int64_t)blockIdx.x * kSplitT + threadIdx.x;
    const int64_t g = row0 + r;
    int cl = 0, cu = 0;
    if (r < n_rows)
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
            const int64_t c = col[k];
            cl += c < g;
            cu += c > g;
        }
    sl[threadIdx.x] = cl;
    su[threadIdx.x] = cu;
    __syncthreads();
    for (int off = 1; off < kSplitT; off <<= 1) {
        int64_t al = 0, au = 0;
        if ((int)threadIdx.x >= off) { al = sl[threadIdx.x - off]; au = su[threadIdx.x - off]; }
        __syncthreads();
        sl[threadIdx.x] += al;
        su[threadIdx.x] += au;
        __syncthreads();
    }
    {
        const int64_t pl = blk_l[blockIdx.x] + sl[threadIdx.x] - cl;
        const int64_t pu = blk_u[blockIdx.x] + su[threadIdx.x] - cu;
        __syncthreads();
        sl[threadIdx.x] = pl; // (from here on: where the row's entries go)
        su[threadIdx.x] = pu;
        if (r < n_rows) {
            rpL[r] = (RPO)pl;
            rpU[r] = (RPO)pu;
            if (r == n_rows - 1) { rpL[n_rows] = (RPO)(pl + cl); rpU[n_rows] = (RPO)(pu + cu); }
        }
        __syncthreads();
    }
    // the entries: a WAVE per row, 64 entries at a time (coalesced reads and writes; a lane walking its row alone moved
    // 1.25 GB in 6.7 ms); positions inside L and U from the ballots, so the order inside a row is kept
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int t = wave; t < kSplitT; t += kSplitT / 64) {
        const int64_t rr = (int64_t)blockIdx.x * kSplitT + t;
        if (rr >= n_rows) break; // (wave-uniform)
        const int64_t gg = row0 + rr;
        int64_t pl = sl[t], pu = su[t];
        bool have_diag = false;
        const int64_t e = (int64_t)row_ptr[rr + 1];
        for (int64_t base = (int64_t)row_ptr[rr]; base < e; base += 64) {
            const int64_t k = base + lane;
            const bool in = k < e;
            const int64_t c = in ? (int64_t)col[k] : gg;
            const double v = in ? val[k] : 0.0;
            const unsigned long long mL = __ballot(in && c < gg), mU = __ballot(in && c > gg), mD = __ballot(in && c == gg);
            if (in && c < gg) { const int64_t q = pl + __popcll(mL & below); colL[q] = (int32_t)c; valL[q] = v; }
            if (in && c > gg) { const int64_t q = pu + __popcll(mU & below); colU[q] = (int32_t)c; valU[q] = v; }
            pl += __popcll(mL);
            pu += __popcll(mU);
            if (mD) { // peel_diag_crs keeps the LAST diagonal entry it meets (:843-857); every one of them is tested
                have_diag = true;
                if (in && c == gg && fabs(v) < 1e-16) atomicMin(status, ((unsigned long long)(gg + 1) << 1) | 0ull);
                if (lane == 63 - __builtin_clzll(mD)) {
                    if (D) D[rr] = v;
                    if (D_inv) D_inv[rr] = 1.0 / v;
                }
            }
        }
        if (!have_diag && lane == 0) atomicMin(status, ((unsigned long long)(gg + 1) << 1) | 1ull);
    }
}

template <typename RP>
bis_status finalize_t(bis_ctx *ctx, bis_mat *A) {
    const RP *rp = (const RP *)A->row_ptr;
    int *d_max = (int *)ctx->counters + 32;
    BIS_HIP_CHECK(ctx, hipMemsetAsync(d_max, 0, sizeof(int), ctx->stream));
    if (A->n_rows > 0) {
        int grid = (int)std::min<int64_t>((A->n_rows + 255) / 256, 4096);
        hipLaunchKernelGGL(max_row_kernel<RP>, dim3(grid), dim3(256), 0, ctx->stream, rp,
                           A->n_rows, d_max);
    }
    int h_max = 0;
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&h_max, d_max, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    A->max_row_nnz = h_max;
    // Row-block tables (0: plain SpMV / triangular sweeps, 1: SpMV with the fused
    // dot epilogue of CG).  With the packed column stream both use one pass of the
    // 256-thread kernel with 2 staged vectors per lane: chunk = 2048 - max_row - 4,
    // so that no block exceeds 2048 stream positions (tools/spmv_ab.py, HPCG-256,
    // same arrays: 0.95 ms against 0.99 at chunk 990).  If some block cannot be
    // packed (more than 8 column windows) the 32-bit stream keeps its own tuning:
    // chunk 1024 plain / 2048 fused (1.03 vs 1.10 ms plain, 1.14 vs 1.11 ms fused).
    const bool user_chunks = bis_opts().spmv_chunk > 0 || bis_opts().spmv_chunk_fused > 0;
    const int packed_mode = bis_opts().spmv_packed < 0 ? 1 : bis_opts().spmv_packed;
    int32_t **rows[2] = {&A->blk_row, &A->blkf_row};
    int64_t **nnzs[2] = {&A->blk_nnz, &A->blkf_nnz};
    int *counts[2] = {&A->n_blocks, &A->n_blocks_f};
    int *chk[2] = {&A->chunk_nnz, &A->chunk_f};
    for (int attempt = 0; attempt < 2; ++attempt) {
        int chunks[2] = {bis_opts().spmv_chunk > 0 ? std::max(256, bis_opts().spmv_chunk) : 1024,
                         bis_opts().spmv_chunk_fused > 0 ? std::max(256, bis_opts().spmv_chunk_fused) : 2048};
        const bool try_single = attempt == 0 && packed_mode != 0 && !user_chunks && A->nnz > 0;
        if (try_single) chunks[0] = chunks[1] = std::max(256, 2048 - h_max - 4);
        bis_mat_free_meta(A);
        for (int t = 0; t < 2; ++t) {
            const int chunk = chunks[t];
            *chk[t] = chunk;
            int64_t nb = (A->nnz + A->n_rows + chunk - 1) / chunk;
            if (nb < 1) nb = 1;
            *counts[t] = (int)nb;
            BIS_HIP_CHECK(ctx, hipMalloc(rows[t], sizeof(int32_t) * (size_t)(nb + 1)));
            BIS_HIP_CHECK(ctx, hipMalloc(nnzs[t], sizeof(int64_t) * (size_t)(nb + 1)));
            hipLaunchKernelGGL(row_blocks_kernel<RP>, dim3((unsigned)((nb + 1 + 255) / 256)), dim3(256), 0,
                               ctx->stream, rp, A->n_rows, (int)nb, (int64_t)chunk, *rows[t], *nnzs[t]);
        }
        BIS_HIP_CHECK(ctx, hipGetLastError());
        if (!try_single) break;
        if (bis_status st = bis_spmv_try_pack(ctx, A, 1)) return st;
        if (A->pk_state[1] == 1) break; // packed: keep the single-pass tables
    }
    return bis_spmv_build_window(ctx, A);
}

} // namespace

// The values of A changed in place (bis_mat_scale_sym; a caller that wrote through bis_mat_debug_ptrs and then calls
// bis_mat_retune): everything derived from them goes -- value dictionary and code streams, the tiled sweeps' plans (they
// hold a copy of the values in their entry stream), the level plans (their row views carry dictionaries of their own).
void bis_mat_values_changed(bis_mat *A) {
    bis_spmv_drop_valdict(A);
    bis_trsv_tiled_destroy(A->tiled_fwd);
    bis_trsv_tiled_destroy(A->tiled_bwd);
    A->tiled_fwd = A->tiled_bwd = nullptr;
    A->tiled_tried_fwd = A->tiled_tried_bwd = false;
    bis_trsv_plan_destroy(A->plan_fwd);
    bis_trsv_plan_destroy(A->plan_bwd);
    A->plan_fwd = A->plan_bwd = nullptr;
}

void bis_mat_free_meta(bis_mat *A) {
    hipFree(A->blk_row); hipFree(A->blk_nnz); hipFree(A->blkf_row); hipFree(A->blkf_nnz);
    A->blk_row = A->blkf_row = nullptr;
    A->blk_nnz = A->blkf_nnz = nullptr;
    bis_spmv_drop_packed(A);
    bis_spmv_drop_valdict(A);
    hipFree(A->loc); hipFree(A->tiles); hipFree(A->tile_cnt);
    A->loc = nullptr; A->tiles = nullptr; A->tile_cnt = nullptr;
    A->win_ok = false;
}

bis_status bis_mat_alloc(bis_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, bool rp64,
                         bis_mat **out) {
    bis_mat *A = new bis_mat;
    A->n_rows = n_rows;
    A->n_cols = n_cols;
    A->nnz = nnz;
    A->rp64 = rp64;
    const size_t rpb = (rp64 ? sizeof(int64_t) : sizeof(int32_t)) * (size_t)(n_rows + 1);
    hipError_t e = hipMalloc(&A->row_ptr, rpb);
    if (e == hipSuccess) e = hipMalloc(&A->col, sizeof(int32_t) * (size_t)(nnz + kPad));
    if (e == hipSuccess) e = hipMalloc(&A->val, sizeof(double) * (size_t)(nnz + kPad));
    if (e == hipSuccess) e = hipMemsetAsync(A->col + nnz, 0, sizeof(int32_t) * kPad, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(A->val + nnz, 0, sizeof(double) * kPad, ctx->stream);
    if (e != hipSuccess) {
        ctx->err = std::string("bis_mat_alloc: ") + hipGetErrorString(e);
        hipFree(A->row_ptr); hipFree(A->col); hipFree(A->val);
        delete A;
        return BIS_ERR_HIP;
    }
    *out = A;
    return BIS_OK;
}

bis_status bis_mat_finalize(bis_ctx *ctx, bis_mat *A) {
    return A->rp64 ? finalize_t<int64_t>(ctx, A) : finalize_t<int32_t>(ctx, A);
}

bis_status bis_mat_row_view(bis_ctx *ctx, const bis_mat *A, int64_t ra, int64_t rb, bis_mat **out) {
    BIS_REQUIRE(ctx, A && out && 0 <= ra && ra <= rb && rb <= A->n_rows, "bis_mat_row_view: bad range");
    int64_t ends[2] = {0, 0};
    const size_t w = A->rp64 ? 8 : 4;
    int64_t a64 = 0, b64 = 0;
    int32_t a32 = 0, b32 = 0;
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(A->rp64 ? (void *)&a64 : (void *)&a32, (char *)A->row_ptr + w * ra, w,
                                      hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(A->rp64 ? (void *)&b64 : (void *)&b32, (char *)A->row_ptr + w * rb, w,
                                      hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ends[0] = A->rp64 ? a64 : a32;
    ends[1] = A->rp64 ? b64 : b32;
    bis_mat *V = new bis_mat;
    V->view = true;
    V->rp64 = A->rp64;
    V->n_rows = rb - ra;
    V->n_cols = A->n_cols;
    V->view_row0 = A->view_row0 + ra;
    V->nnz = ends[1] - ends[0];
    V->row_ptr = (char *)A->row_ptr + w * ra;
    V->col = A->col;
    V->val = A->val;
    bis_status st = bis_mat_finalize(ctx, V);
    if (st != BIS_OK) { delete V; return st; }
    *out = V;
    return BIS_OK;
}

// Does the host CRS look like a one-unknown-per-node stencil on an nx x ny x nz grid in natural order (x fastest)?  Read
// off the positive column offsets of a few rows in the middle of the matrix: {1}, a run around nx, runs around the
// plane size nx*ny.  Only a HINT for the tiled triangular sweep, which verifies the order it derives from it against
// every entry (bis_trsv_tiled.hip): a wrong guess costs one refused plan, never a wrong result.
template <class RP>
static bool guess_grid_hint(int64_t n, const RP *rp, const int32_t *col, int64_t g[4]) {
    if (n < 4096) return false;
    int64_t last[3] = {0, 0, 0};
    int agree = 0;
    for (int attempt = 0; attempt < 24 && agree < 2; ++attempt) {
        const int64_t r = (n / 2 + (int64_t)attempt * 37) % n;
        std::vector<int64_t> pos;
        for (int64_t k = (int64_t)rp[r]; k < (int64_t)rp[r + 1]; ++k)
            if (col[k] > r) pos.push_back((int64_t)col[k] - r);
        std::sort(pos.begin(), pos.end());
        pos.erase(std::unique(pos.begin(), pos.end()), pos.end());
        if (pos.empty() || pos[0] != 1) continue;
        std::vector<std::pair<int64_t, int64_t>> runs; // maximal runs of consecutive offsets
        for (int64_t o : pos) {
            if (!runs.empty() && o == runs.back().second + 1) runs.back().second = o;
            else runs.push_back({o, o});
        }
        if (runs[0] != std::make_pair<int64_t, int64_t>(1, 1) || runs.size() < 2) continue;
        bool ok = true;
        std::vector<int64_t> centre;
        for (size_t i = 1; i < runs.size() && ok; ++i) {
            const int64_t w = runs[i].second - runs[i].first + 1;
            if (w == 1) centre.push_back(runs[i].first);
            else if (w == 3) centre.push_back(runs[i].first + 1);
            else ok = false;
        }
        if (!ok) continue;
        const int64_t nx = centre[0];
        int64_t plane = 0;
        if (centre.size() >= 4 && centre[2] - centre[1] == nx && centre[3] - centre[2] == nx) plane = centre[2];
        else if (centre.size() >= 2) plane = centre[1];
        int64_t ny, nz;
        if (plane == 0) {
            if (n % nx) continue;
            ny = n / nx; nz = 1;
        } else {
            if (plane % nx || n % plane) continue;
            ny = plane / nx; nz = n / plane;
        }
        if (nx < 2 || ny < 2) continue;
        if (agree && last[0] == nx && last[1] == ny && last[2] == nz) ++agree;
        else { last[0] = nx; last[1] = ny; last[2] = nz; agree = 1; }
    }
    if (agree < 2) return false;
    g[0] = last[0]; g[1] = last[1]; g[2] = last[2]; g[3] = 1;
    return true;
}

extern "C" {

static bis_status mat_create_common(bis_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz,
                                    const void *row_ptr, bool src64, const int32_t *col,
                                    const double *val, bis_mat **out) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, out && n_rows >= 0 && n_cols >= 0 && nnz >= 0 && row_ptr &&
                         (nnz == 0 || (col && val)),
                "bis_mat_create: bad arguments");
    BIS_REQUIRE(ctx, n_rows < INT32_MAX && n_cols < INT32_MAX,
                "bis_mat_create: row/column count must fit int32 (col is int32)");
    const bool rp64 = bis_want_rp64(nnz);
    bis_mat *A = nullptr;
    bis_status st = bis_mat_alloc(ctx, n_rows, n_cols, nnz, rp64, &A);
    if (st != BIS_OK) return st;
    // convert row_ptr width on the host if source and device widths differ
    std::vector<int64_t> tmp64;
    std::vector<int32_t> tmp32;
    const void *src = row_ptr;
    if (src64 && !rp64) {
        tmp32.resize(n_rows + 1);
        for (int64_t i = 0; i <= n_rows; ++i) tmp32[i] = (int32_t)((const int64_t *)row_ptr)[i];
        src = tmp32.data();
    } else if (!src64 && rp64) {
        tmp64.resize(n_rows + 1);
        for (int64_t i = 0; i <= n_rows; ++i) tmp64[i] = ((const int32_t *)row_ptr)[i];
        src = tmp64.data();
    }
    hipError_t e = hipMemcpyAsync(A->row_ptr, src, (rp64 ? 8 : 4) * (size_t)(n_rows + 1),
                                  hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && nnz)
        e = hipMemcpyAsync(A->col, col, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && nnz)
        e = hipMemcpyAsync(A->val, val, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ctx->err = std::string("bis_mat_create: ") + hipGetErrorString(e);
        bis_mat_destroy(ctx, A);
        return BIS_ERR_HIP;
    }
    // a malformed input must be an error here, not an out-of-bounds gather in a kernel later
    if (n_rows > 0) {
        int *status = (int *)ctx->counters + 36;
        hipMemsetAsync(status, 0, sizeof(int), ctx->stream);
        const unsigned grid = (unsigned)((n_rows + 255) / 256);
        if (rp64) hipLaunchKernelGGL(validate_kernel<int64_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int64_t *)A->row_ptr, A->col, n_rows, n_cols, nnz, status);
        else hipLaunchKernelGGL(validate_kernel<int32_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col, n_rows, n_cols, nnz, status);
        int h = 0;
        e = hipMemcpyAsync(&h, status, sizeof h, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ctx->err = std::string("bis_mat_create: ") + hipGetErrorString(e); bis_mat_destroy(ctx, A); return BIS_ERR_HIP; }
        if (h) {
            ctx->err = (h & 1) ? "bis_mat_create: row_ptr is not a monotone sequence from 0 to nnz"
                               : "bis_mat_create: column index outside [0, n_cols)";
            bis_mat_destroy(ctx, A);
            return BIS_ERR_INVALID;
        }
    } else if (nnz != 0) {
        ctx->err = "bis_mat_create: non-zeros without rows";
        bis_mat_destroy(ctx, A);
        return BIS_ERR_INVALID;
    }
    if (bis_opts().grid_autodetect != 0 && n_rows == n_cols && nnz > 0) {
        int64_t g[4];
        const bool hit = src64 ? guess_grid_hint(n_rows, (const int64_t *)row_ptr, col, g) : guess_grid_hint(n_rows, (const int32_t *)row_ptr, col, g);
        if (hit) for (int i = 0; i < 4; ++i) A->grid[i] = g[i];
    }
    st = bis_mat_finalize(ctx, A);
    if (st != BIS_OK) { bis_mat_destroy(ctx, A); return st; }
    *out = A;
    return BIS_OK;
}

bis_status bis_mat_create(bis_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz,
                          const int32_t *row_ptr, const int32_t *col, const double *val,
                          bis_mat **out) {
    return mat_create_common(ctx, n_rows, n_cols, nnz, row_ptr, false, col, val, out);
}

bis_status bis_mat_create64(bis_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz,
                            const int64_t *row_ptr, const int32_t *col, const double *val,
                            bis_mat **out) {
    return mat_create_common(ctx, n_rows, n_cols, nnz, row_ptr, true, col, val, out);
}

bis_status bis_mat_destroy(bis_ctx *ctx, bis_mat *A) {
    BIS_CTX_OK(ctx);
    if (!A) return BIS_OK;
    hipStreamSynchronize(ctx->stream);
    bis_trsv_plan_destroy(A->plan_fwd);
    bis_trsv_plan_destroy(A->plan_bwd);
    bis_trsv_tiled_destroy(A->tiled_fwd);
    bis_trsv_tiled_destroy(A->tiled_bwd);
    bis_trsv_chain_destroy(A->chain_fwd);
    bis_trsv_chain_destroy(A->chain_bwd);
    if (!A->view) {
        hipFree(A->row_ptr);
        hipFree(A->col);
        hipFree(A->val);
    }
    bis_mat_free_meta(A);
    delete A;
    return BIS_OK;
}

// tuning aid: rebuild the row-block metadata with the current options
BIS_API bis_status bis_mat_retune(bis_ctx *ctx, bis_mat *A) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A, "bis_mat_retune: null matrix");
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    bis_mat_values_changed(A); // (the tuning tools may also have written the arrays through bis_mat_debug_ptrs)
    return bis_mat_finalize(ctx, A);
}

// Placement tuning: see include/bis_hip.h.
BIS_API bis_status bis_mat_tune_placement(bis_ctx *ctx, bis_mat *A, int max_trials, double *first_ms,
                                          double *best_ms) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A && !A->view, "bis_mat_tune_placement: owning matrix required");
    // the streamed arrays are re-allocated: row views made earlier -- the triangular-solve plans cache some -- would
    // keep pointing at the freed ones
    BIS_REQUIRE(ctx, !A->plan_fwd && !A->plan_bwd && !A->tiled_fwd && !A->tiled_bwd && !A->chain_fwd && !A->chain_bwd, "bis_mat_tune_placement: call it before the first triangular solve on this matrix");
    if (first_ms) *first_ms = 0.0;
    if (best_ms) *best_ms = 0.0;
    if (A->nnz == 0 || A->n_rows == 0 || max_trials <= 0) return BIS_OK;
    double *x = nullptr, *y = nullptr;
    bis_status st = bis_vec_alloc(ctx, A->n_cols, &x);
    if (st == BIS_OK) st = bis_vec_alloc(ctx, A->n_rows, &y);
    if (st == BIS_OK) st = bis_init_vector(ctx, x, 1.0, A->n_cols);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (st == BIS_OK && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) st = BIS_ERR_HIP;
    auto measure = [&](double &ms) -> bis_status {
        for (int i = 0; i < 2; ++i) { bis_status s2 = bis_spmv(ctx, A, x, y); if (s2 != BIS_OK) return s2; }
        hipEventRecord(e0, ctx->stream);
        for (int i = 0; i < 5; ++i) { bis_status s2 = bis_spmv(ctx, A, x, y); if (s2 != BIS_OK) return s2; }
        hipEventRecord(e1, ctx->stream);
        if (hipEventSynchronize(e1) != hipSuccess) return BIS_ERR_HIP;
        float f = 0.f;
        hipEventElapsedTime(&f, e0, e1);
        ms = f / 5.0;
        return BIS_OK;
    };
    // the streamed arrays: values, and the packed column stream (or the 32-bit columns)
    struct Set { double *val; int32_t *col; uint16_t *pk; };
    const int t = 1; // identical tables share stream 1 (bis_spmv.hip ensure_packed)
    const bool packed = A->pk_state[t] == 1 && A->chunk_nnz == A->chunk_f;
    const size_t n_val = (size_t)A->nnz + 8; // kPad
    size_t n_pk = 0;
    if (packed) {
        int64_t ends[2];
        hipMemcpyAsync(&ends[0], A->blkf_nnz, 8, hipMemcpyDeviceToHost, ctx->stream);
        hipMemcpyAsync(&ends[1], A->blkf_nnz + A->n_blocks_f, 8, hipMemcpyDeviceToHost, ctx->stream);
        hipStreamSynchronize(ctx->stream);
        n_pk = (size_t)(ends[1] - A->pk_base[t]) + 16;
    }
    std::vector<Set> rejected;
    double best = 0.0;
    if (st == BIS_OK) st = measure(best);
    if (first_ms) *first_ms = best;
    // The window + sliced-ELL form with the 8-byte values (win8) reads ONE array of its own, and where that array lies decides
    // between two levels of the kernel's time, 13 % apart (HPCG-256: 0.755 / 0.855 ms, constant over time for an allocation,
    // independent of where x and y lie; tools/win8_place2.py, profiles/r05_f_win8_placement.log): candidates are copies of the
    // stream in fresh allocations, the earlier ones held so that the next lands elsewhere; the search ends at the first
    // candidate the kernel reads at >= 5.9 TB/s (the fast level: 6.0-6.5; the library runs the same search by itself when it
    // builds the stream -- bis_spmv_sell.hip w8_tune_placement -- so this normally finds the fast level in place).
    if (st == BIS_OK && A->sw8_state == 1 && bis_spmv_win8_stream_bytes(A) > 0) {
        int w8_form = 0;
        if (bis_mat_spmv_stream_info(ctx, A, nullptr, nullptr, nullptr, &w8_form) == BIS_OK && w8_form == 6) {
            const size_t bytes = bis_spmv_win8_stream_bytes(A);
            std::vector<void *> losers;
            auto fast_enough = [&](double ms) { return (double)bytes / (ms * 1e-3) >= 5.9e12; };
            for (int trial = 0; st == BIS_OK && trial < max_trials && !fast_enough(best); ++trial) {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + ((size_t)4 << 30)) break; // keep 4 GiB for the caller
                void *cand = nullptr;
                if (hipMalloc(&cand, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
                void *cur = bis_spmv_win8_swap_stream(A, cand);
                hipMemcpyAsync(cand, cur, bytes, hipMemcpyDeviceToDevice, ctx->stream);
                double ms = 0.0;
                st = measure(ms);
                if (st == BIS_OK && ms < best) { best = ms; losers.push_back(cur); }
                else { bis_spmv_win8_swap_stream(A, cur); losers.push_back(cand); }
            }
            hipStreamSynchronize(ctx->stream);
            for (void *l : losers) hipFree(l);
            if (best_ms) *best_ms = best;
            if (e0) hipEventDestroy(e0);
            if (e1) hipEventDestroy(e1);
            bis_vec_free(ctx, x);
            bis_vec_free(ctx, y);
            return st;
        }
    }
    // the value-dictionary kernels stream a quarter of the bytes and are not HBM-bound: where their arrays lie matters
    // little, and the arrays re-allocated below (values, packed columns of the row-block tables) are not the ones they read
    if (A->vd_state == 1) max_trials = 0;
    for (int trial = 0; st == BIS_OK && trial < max_trials; ++trial) {
        Set cand{nullptr, nullptr, nullptr};
        hipError_t e = hipMalloc(&cand.val, sizeof(double) * n_val);
        if (e == hipSuccess && packed) e = hipMalloc(&cand.pk, sizeof(uint16_t) * n_pk);
        if (e == hipSuccess && !packed) e = hipMalloc(&cand.col, sizeof(int32_t) * n_val);
        if (e != hipSuccess) { // out of memory: stop trying, keep what we have
            hipFree(cand.val); hipFree(cand.pk); hipFree(cand.col);
            (void)hipGetLastError();
            break;
        }
        hipMemcpyAsync(cand.val, A->val, sizeof(double) * n_val, hipMemcpyDeviceToDevice, ctx->stream);
        if (packed) hipMemcpyAsync(cand.pk, A->pk[t], sizeof(uint16_t) * n_pk, hipMemcpyDeviceToDevice, ctx->stream);
        else hipMemcpyAsync(cand.col, A->col, sizeof(int32_t) * n_val, hipMemcpyDeviceToDevice, ctx->stream);
        Set cur{A->val, packed ? nullptr : A->col, packed ? A->pk[t] : nullptr};
        A->val = cand.val;
        if (packed) A->pk[t] = cand.pk; else A->col = cand.col;
        double ms = 0.0;
        st = measure(ms);
        if (st == BIS_OK && ms < best) {
            best = ms;
            rejected.push_back(cur);
        } else {
            A->val = cur.val;
            if (packed) A->pk[t] = cur.pk; else A->col = cur.col;
            rejected.push_back(cand);
        }
    }
    hipStreamSynchronize(ctx->stream);
    for (Set &r : rejected) { hipFree(r.val); hipFree(r.col); hipFree(r.pk); }
    if (best_ms) *best_ms = best;
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    bis_vec_free(ctx, x);
    bis_vec_free(ctx, y);
    return st;
}

// debugging / tuning aid: device addresses of the CRS arrays
BIS_API bis_status bis_mat_debug_ptrs(const bis_mat *A, void **row_ptr, void **col, void **val) {
    if (!A) return BIS_ERR_INVALID;
    if (row_ptr) *row_ptr = A->row_ptr;
    if (col) *col = A->col;
    if (val) *val = A->val;
    return BIS_OK;
}

int bis_mat_rp_width(const bis_mat *A) { return A ? (A->rp64 ? 8 : 4) : 0; }

bis_status bis_mat_set_grid_hint(bis_mat *A, int64_t nx, int64_t ny, int64_t nz, int dof) {
    if (!A) return BIS_ERR_INVALID;
    if (nx <= 0 || ny <= 0 || nz <= 0 || dof <= 0 || nx * ny * nz * dof != A->n_rows) return BIS_ERR_INVALID;
    A->grid[0] = nx; A->grid[1] = ny; A->grid[2] = nz; A->grid[3] = dof;
    return BIS_OK;
}

bis_status bis_mat_info(const bis_mat *A, int64_t *n_rows, int64_t *n_cols, int64_t *nnz) {
    if (!A) return BIS_ERR_INVALID;
    if (n_rows) *n_rows = A->n_rows;
    if (n_cols) *n_cols = A->n_cols;
    if (nnz) *nnz = A->nnz;
    return BIS_OK;
}

bis_status bis_mat_download(bis_ctx *ctx, const bis_mat *A, int64_t *row_ptr, int32_t *col,
                            double *val) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A, "bis_mat_download: null matrix");
    if (row_ptr) {
        if (A->rp64) {
            BIS_HIP_CHECK(ctx, hipMemcpyAsync(row_ptr, A->row_ptr, 8 * (size_t)(A->n_rows + 1),
                                              hipMemcpyDeviceToHost, ctx->stream));
            BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        } else {
            std::vector<int32_t> t(A->n_rows + 1);
            BIS_HIP_CHECK(ctx, hipMemcpyAsync(t.data(), A->row_ptr, 4 * (size_t)(A->n_rows + 1),
                                              hipMemcpyDeviceToHost, ctx->stream));
            BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            for (int64_t i = 0; i <= A->n_rows; ++i) row_ptr[i] = t[i];
        }
    }
    if (col && A->nnz)
        BIS_HIP_CHECK(ctx, hipMemcpyAsync(col, A->col, 4 * (size_t)A->nnz, hipMemcpyDeviceToHost, ctx->stream));
    if (val && A->nnz)
        BIS_HIP_CHECK(ctx, hipMemcpyAsync(val, A->val, 8 * (size_t)A->nnz, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return BIS_OK;
}

bis_status bis_mat_gen_hpcg(bis_ctx *ctx, int64_t nx, int64_t ny, int64_t nz, int64_t row0,
                            int64_t row1, bis_mat **out) {
    BIS_CTX_OK(ctx);
    const int64_t N = nx * ny * nz;
    BIS_REQUIRE(ctx, out && nx > 0 && ny > 0 && nz > 0 && N < INT32_MAX && row0 >= 0 &&
                         row0 <= row1 && row1 <= N,
                "bis_mat_gen_hpcg: bad arguments");
    const int64_t base = hpcg_row_ptr_host(row0, nx, ny, nz);
    const int64_t nnz = hpcg_row_ptr_host(row1, nx, ny, nz) - base;
    const int64_t n_local = row1 - row0;
    const bool rp64 = bis_want_rp64(nnz);
    bis_mat *A = nullptr;
    bis_status st = bis_mat_alloc(ctx, n_local, N, nnz, rp64, &A);
    if (st != BIS_OK) return st;
    const unsigned grid = (unsigned)((n_local + 1 + 255) / 256);
    if (rp64)
        hipLaunchKernelGGL(gen_hpcg_kernel<int64_t>, dim3(grid), dim3(256), 0, ctx->stream, nx, ny,
                           nz, row0, row1, base, (int64_t *)A->row_ptr, A->col, A->val);
    else
        hipLaunchKernelGGL(gen_hpcg_kernel<int32_t>, dim3(grid), dim3(256), 0, ctx->stream, nx, ny,
                           nz, row0, row1, base, (int32_t *)A->row_ptr, A->col, A->val);
    if (hipGetLastError() != hipSuccess) { bis_mat_destroy(ctx, A); ctx->err = "gen_hpcg launch failed"; return BIS_ERR_HIP; }
    if (row0 == 0 && row1 == N) { A->grid[0] = nx; A->grid[1] = ny; A->grid[2] = nz; A->grid[3] = 1; }
    st = bis_mat_finalize(ctx, A);
    if (st != BIS_OK) { bis_mat_destroy(ctx, A); return st; }
    *out = A;
    return BIS_OK;
}

bis_status bis_mat_gen_anderson(bis_ctx *ctx, int64_t L, double t, double W, double shift,
                                uint64_t seed, int64_t row0, int64_t row1, bis_mat **out) {
    BIS_CTX_OK(ctx);
    const int64_t N = L * L * L;
    BIS_REQUIRE(ctx, out && L >= 3 && N < INT32_MAX && row0 >= 0 && row0 <= row1 && row1 <= N,
                "bis_mat_gen_anderson: bad arguments (L >= 3)");
    const int64_t n_local = row1 - row0, nnz = 7 * n_local;
    const bool rp64 = bis_want_rp64(nnz);
    bis_mat *A = nullptr;
    bis_status st = bis_mat_alloc(ctx, n_local, N, nnz, rp64, &A);
    if (st != BIS_OK) return st;
    const unsigned grid = (unsigned)((n_local + 1 + 255) / 256);
    if (rp64)
        hipLaunchKernelGGL(gen_anderson_kernel<int64_t>, dim3(grid), dim3(256), 0, ctx->stream, L, t,
                           W, shift, seed, row0, row1, (int64_t *)A->row_ptr, A->col, A->val);
    else
        hipLaunchKernelGGL(gen_anderson_kernel<int32_t>, dim3(grid), dim3(256), 0, ctx->stream, L, t,
                           W, shift, seed, row0, row1, (int32_t *)A->row_ptr, A->col, A->val);
    if (hipGetLastError() != hipSuccess) { bis_mat_destroy(ctx, A); ctx->err = "gen_anderson launch failed"; return BIS_ERR_HIP; }
    if (row0 == 0 && row1 == N) { A->grid[0] = L; A->grid[1] = L; A->grid[2] = L; A->grid[3] = 1; }
    st = bis_mat_finalize(ctx, A);
    if (st != BIS_OK) { bis_mat_destroy(ctx, A); return st; }
    *out = A;
    return BIS_OK;
}

bis_status bis_mat_gen_fem(bis_ctx *ctx, int64_t nx, int64_t ny, int64_t nz, int keep_percent,
                           uint64_t seed, int64_t row0, int64_t row1, bis_mat **out) {
    BIS_CTX_OK(ctx);
    const int64_t N = 3 * nx * ny * nz;
    BIS_REQUIRE(ctx, out && nx >= 1 && ny >= 1 && nz >= 1 && N < INT32_MAX && keep_percent >= 0 &&
                         keep_percent <= 100 && row0 >= 0 && row0 <= row1 && row1 <= N,
                "bis_mat_gen_fem: bad arguments");
    const int64_t n_local = row1 - row0;
    const int n_blk = (int)((n_local + 255) / 256);
    int64_t *blk = nullptr;
    BIS_HIP_CHECK(ctx, hipMalloc(&blk, sizeof(int64_t) * (size_t)(2 * n_blk + 4)));
    int64_t *dummy = blk + n_blk, *tot = blk + 2 * n_blk;
    BIS_HIP_CHECK(ctx, hipMemsetAsync(blk, 0, sizeof(int64_t) * (size_t)(2 * n_blk + 4), ctx->stream));
    if (n_blk > 0) {
        hipLaunchKernelGGL(fem_count_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, nx, ny, nz, keep_percent,
                           seed, row0, n_local, blk);
        hipLaunchKernelGGL(split_scan_kernel, dim3(1), dim3(256), 0, ctx->stream, blk, dummy, n_blk, tot);
    }
    int64_t h_tot[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(h_tot, tot, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { hipFree(blk); ctx->err = "bis_mat_gen_fem: count pass failed"; return BIS_ERR_HIP; }
    const int64_t nnz = h_tot[0];
    const bool rp64 = bis_want_rp64(nnz);
    bis_mat *A = nullptr;
    bis_status st = bis_mat_alloc(ctx, n_local, N, nnz, rp64, &A);
    if (st != BIS_OK) { hipFree(blk); return st; }
    if (n_blk > 0) {
        if (rp64)
            hipLaunchKernelGGL(fem_fill_kernel<int64_t>, dim3(n_blk), dim3(256), 0, ctx->stream, nx, ny, nz,
                               keep_percent, seed, row0, n_local, blk, (int64_t *)A->row_ptr, A->col, A->val);
        else
            hipLaunchKernelGGL(fem_fill_kernel<int32_t>, dim3(n_blk), dim3(256), 0, ctx->stream, nx, ny, nz,
                               keep_percent, seed, row0, n_local, blk, (int32_t *)A->row_ptr, A->col, A->val);
    } else {
        hipMemsetAsync(A->row_ptr, 0, rp64 ? 8 : 4, ctx->stream);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(blk);
    if (e != hipSuccess) { bis_mat_destroy(ctx, A); ctx->err = "gen_fem launch failed"; return BIS_ERR_HIP; }
    if (row0 == 0 && row1 == N) { A->grid[0] = nx; A->grid[1] = ny; A->grid[2] = nz; A->grid[3] = 3; }
    st = bis_mat_finalize(ctx, A);
    if (st != BIS_OK) { bis_mat_destroy(ctx, A); return st; }
    *out = A;
    return BIS_OK;
}

} // extern "C"

bis_status bis_mat_split_strict_impl(bis_ctx *ctx, const bis_mat *A, bis_mat **L_strict,
                                     bis_mat **U_strict, double *D, double *D_inv, bool check_diag);

extern "C" {

bis_status bis_mat_split_strict(bis_ctx *ctx, const bis_mat *A, bis_mat **L_strict,
                                bis_mat **U_strict, double *D, double *D_inv) {
    return bis_mat_split_strict_impl(ctx, A, L_strict, U_strict, D, D_inv, true);
}

} // extern "C"

bis_status bis_mat_split_strict_impl(bis_ctx *ctx, const bis_mat *A, bis_mat **L_strict,
                                     bis_mat **U_strict, double *D, double *D_inv, bool check_diag) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A && L_strict && U_strict, "bis_mat_split_strict: bad arguments");
    BIS_REQUIRE(ctx, A->n_rows == A->n_cols, "bis_mat_split_strict: square local matrix required");
    const int64_t n = A->n_rows;
    const int n_blk = (int)((n + kSplitT - 1) / kSplitT);
    int64_t *blk = nullptr;
    unsigned long long *status = nullptr;
    BIS_HIP_CHECK(ctx, hipMalloc(&blk, sizeof(int64_t) * (size_t)(2 * n_blk + 2) + 16));
    int64_t *blk_l = blk, *blk_u = blk + n_blk, *tot = blk + 2 * n_blk;
    status = (unsigned long long *)(ctx->scalars_dev + 32);
    BIS_HIP_CHECK(ctx, hipMemsetAsync(status, 0xFF, 8, ctx->stream));
    BIS_HIP_CHECK(ctx, hipMemsetAsync(tot, 0, 16, ctx->stream));
    if (n_blk > 0) {
        if (A->rp64)
            hipLaunchKernelGGL(split_count_kernel<int64_t>, dim3(n_blk), dim3(kSplitT), 0, ctx->stream,
                               (const int64_t *)A->row_ptr, A->col, n, (int64_t)0, blk_l, blk_u);
        else
            hipLaunchKernelGGL(split_count_kernel<int32_t>, dim3(n_blk), dim3(kSplitT), 0, ctx->stream,
                               (const int32_t *)A->row_ptr, A->col, n, (int64_t)0, blk_l, blk_u);
        hipLaunchKernelGGL(split_scan_kernel, dim3(1), dim3(256), 0, ctx->stream, blk_l, blk_u, n_blk, tot);
    }
    int64_t h_tot[2] = {0, 0};
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(h_tot, tot, 16, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    bis_mat *Lm = nullptr, *Um = nullptr;
    // strict parts of a matrix whose nnz fits int32 also fit
    const bool rpo64 = A->rp64 && (bis_want_rp64(h_tot[0]) || bis_want_rp64(h_tot[1]));
    bis_status st = bis_mat_alloc(ctx, n, n, h_tot[0], rpo64, &Lm);
    if (st == BIS_OK) st = bis_mat_alloc(ctx, n, n, h_tot[1], rpo64, &Um);
    if (st != BIS_OK) { hipFree(blk); if (Lm) bis_mat_destroy(ctx, Lm); return st; }
    if (n_blk > 0) {
#define BIS_SPLIT_FILL(RP, RPO)                                                                  \
    hipLaunchKernelGGL((split_fill_kernel<RP, RPO>), dim3(n_blk), dim3(kSplitT), 0, ctx->stream, \
                       (const RP *)A->row_ptr, A->col, A->val, n, (int64_t)0, blk_l, blk_u,      \
                       (RPO *)Lm->row_ptr, Lm->col, Lm->val, (RPO *)Um->row_ptr, Um->col,        \
                       Um->val, D, D_inv, status)
        if (A->rp64 && rpo64) BIS_SPLIT_FILL(int64_t, int64_t);
        else if (A->rp64) BIS_SPLIT_FILL(int64_t, int32_t);
        else BIS_SPLIT_FILL(int32_t, int32_t);
#undef BIS_SPLIT_FILL
    } else {
        hipMemsetAsync(Lm->row_ptr, 0, rpo64 ? 8 : 4, ctx->stream);
        hipMemsetAsync(Um->row_ptr, 0, rpo64 ? 8 : 4, ctx->stream);
    }
    unsigned long long h_status = 0;
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&h_status, status, 8, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    hipFree(blk);
    if (check_diag && h_status != ~0ull) {
        const long long row = (long long)(h_status >> 1) - 1;
        const bool missing = h_status & 1ull;
        char msg[128];
        // same texts as SanityChecker::zero_diag / no_diag (common.hpp:388-396)
        snprintf(msg, sizeof msg, missing ? "No diagonal to extract at row index %lld"
                                          : "Zero detected on diagonal at row index %lld", row);
        ctx->err = msg;
        bis_mat_destroy(ctx, Lm);
        bis_mat_destroy(ctx, Um);
        return missing ? BIS_ERR_NO_DIAG : BIS_ERR_ZERO_DIAG;
    }
    st = bis_mat_finalize(ctx, Lm);
    if (st == BIS_OK) st = bis_mat_finalize(ctx, Um);
    if (st != BIS_OK) { bis_mat_destroy(ctx, Lm); bis_mat_destroy(ctx, Um); return st; }
    for (int i = 0; i < 4; ++i) { Lm->grid[i] = A->grid[i]; Um->grid[i] = A->grid[i]; } // same row space
    *L_strict = Lm;
    *U_strict = Um;
    return BIS_OK;
}
