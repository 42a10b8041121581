// bis_spmv.hip -- CRS SpMV for gfx950 (reference kernels.hpp:22-52).
//
// HBM-bound: algorithmic traffic 12*nnz + 20*N bytes (val 8 + col 4 per
// non-zero; row_ptr 4, x 8, y 8 per row).  Design ("row-block stream"):
//
//   * rows are grouped into row blocks of ~chunk non-zeros (bis_matrix.hip,
//     blk_row[]); one workgroup streams one block's contiguous val/col range
//     with 16-byte vector loads (4 non-zeros per lane per step) -- perfectly
//     coalesced regardless of row lengths -- gathers x[col] through L1/L2,
//     and parks the products in LDS;
//   * after one barrier each lane owns a row and sums that row's products
//     from LDS left to right (CRS storage order, like the reference's scalar
//     loop), then writes y coalesced;
//   * the blockIdx -> row-block map is XCD-aware: each XCD sweeps one
//     contiguous slab of rows so the x planes a stencil row touches stay in
//     that XCD's 4 MiB L2 (MI355X_MICROARCH.md, "Workgroup dispatch").
//
// Rows longer than the LDS budget fall back to a wave-per-row kernel.
// An optional fused epilogue accumulates sum_r y[r]*w[r] (the (Ap,p) of
// cg.hpp:23) into per-block partials so CG needs no separate dot pass.
#include "bis_internal.hpp"

#include <cstdlib>

namespace {

constexpr int kSpmvT = 256;

template <typename RP, bool FUSE_DOT>
__global__ __launch_bounds__(kSpmvT) void spmv_rowblock_kernel(
    const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const int32_t *__restrict__ blk_row, int n_blocks, int n_blocks_pad8,
    const double *__restrict__ w, double *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) double prod[];
    const int b = xcd_remap(blockIdx.x, n_blocks_pad8);
    if (b >= n_blocks) return;
    const int r0 = blk_row[b], r1 = blk_row[b + 1];
    const int64_t s = (int64_t)row_ptr[r0], e = (int64_t)row_ptr[r1];
    const int64_t s4 = s & ~(int64_t)3;

    // phase 1: stream val/col, gather x, park products
    for (int64_t k = s4 + 4 * (int64_t)threadIdx.x; k < e; k += 4 * kSpmvT) {
        const int4 c = *reinterpret_cast<const int4 *>(col + k);
        const double2 v01 = *reinterpret_cast<const double2 *>(val + k);
        const double2 v23 = *reinterpret_cast<const double2 *>(val + k + 2);
        double2 p01, p23;
        p01.x = v01.x * x[c.x];
        p01.y = v01.y * x[c.y];
        p23.x = v23.x * x[c.z];
        p23.y = v23.y * x[c.w];
        double2 *dst = reinterpret_cast<double2 *>(prod + (k - s4));
        dst[0] = p01;
        dst[1] = p23;
    }
    __syncthreads();

    // phase 2: one lane per row, left-to-right sum in CRS order
    double dot_acc = 0.0;
    for (int r = r0 + (int)threadIdx.x; r < r1; r += kSpmvT) {
        const int a = (int)((int64_t)row_ptr[r] - s4), z = (int)((int64_t)row_ptr[r + 1] - s4);
        double acc = 0.0;
        for (int j = a; j < z; ++j) acc += prod[j];
        y[r] = acc;
        if (FUSE_DOT) dot_acc = fma(acc, w[r], dot_acc);
    }
    if (FUSE_DOT) {
        __shared__ double red[kSpmvT / 64];
        const double t = block_sum<kSpmvT>(dot_acc, red);
        if (threadIdx.x == 0) partials[b] = t;
    }
}

// Fallback for rows longer than the LDS budget: one wave per row.
template <typename RP>
__global__ __launch_bounds__(256) void spmv_wave_per_row_kernel(
    const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    int64_t n_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        double acc = 0.0;
        for (int64_t k = (int64_t)row_ptr[r] + lane; k < (int64_t)row_ptr[r + 1]; k += 64)
            acc = fma(val[k], x[col[k]], acc);
        acc = wave_sum(acc);
        if (lane == 0) y[r] = acc;
    }
}

} // namespace

// internal: y = A x, optionally partials[b] = sum_{r in block b} y[r]*w[r]
// (n_partials returns the number of partials written; 0 if not fused).
bis_status bis_spmv_launch(bis_ctx *ctx, const bis_mat *A, const double *x, double *y,
                           const double *w, int *n_partials) {
    if (n_partials) *n_partials = 0;
    if (A->n_rows == 0) return BIS_OK;
    const int64_t lds_doubles = (int64_t)A->chunk_nnz + A->max_row_nnz + 8;
    const size_t lds_bytes = sizeof(double) * (size_t)lds_doubles;
    if (lds_bytes > 64 * 1024) {
        if (w) { ctx->err = "bis_spmv: fused dot unsupported for very long rows"; return BIS_ERR_UNSUPPORTED; }
        const int grid = (int)std::min<int64_t>((A->n_rows + 3) / 4, 8192);
        bis_prof_begin(ctx);
        if (A->rp64)
            hipLaunchKernelGGL(spmv_wave_per_row_kernel<int64_t>, dim3(grid), dim3(256), 0, ctx->stream,
                               (const int64_t *)A->row_ptr, A->col, A->val, x, y, A->n_rows);
        else
            hipLaunchKernelGGL(spmv_wave_per_row_kernel<int32_t>, dim3(grid), dim3(256), 0, ctx->stream,
                               (const int32_t *)A->row_ptr, A->col, A->val, x, y, A->n_rows);
        bis_prof_end(ctx);
        BIS_HIP_CHECK(ctx, hipGetLastError());
        return BIS_OK;
    }
    const int nb = A->n_blocks, nb8 = (nb + 7) & ~7;
    if (w) {
        bis_status st = bis_ensure_partials(ctx, (size_t)nb);
        if (st != BIS_OK) return st;
    }
    bis_prof_begin(ctx);
#define BIS_SPMV_LAUNCH(RP, FUSE)                                                                \
    hipLaunchKernelGGL((spmv_rowblock_kernel<RP, FUSE>), dim3(nb8), dim3(kSpmvT), lds_bytes,     \
                       ctx->stream, (const RP *)A->row_ptr, A->col, A->val, x, y, A->blk_row, nb, \
                       nb8, w, ctx->partials)
    if (A->rp64) { if (w) BIS_SPMV_LAUNCH(int64_t, true); else BIS_SPMV_LAUNCH(int64_t, false); }
    else { if (w) BIS_SPMV_LAUNCH(int32_t, true); else BIS_SPMV_LAUNCH(int32_t, false); }
#undef BIS_SPMV_LAUNCH
    bis_prof_end(ctx);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    if (w && n_partials) *n_partials = nb;
    return BIS_OK;
}

extern "C" {

bis_status bis_spmv(bis_ctx *ctx, const bis_mat *A, const double *x, double *y) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A && (A->n_rows == 0 || (x && y)), "bis_spmv: bad arguments");
    BIS_REQUIRE(ctx, x != y, "bis_spmv: x and y must not alias");
    return bis_spmv_launch(ctx, A, x, y, nullptr, nullptr);
}

bis_status bis_compute_residual(bis_ctx *ctx, const bis_mat *A, const double *x, const double *b,
                                double *res, double *tmp) {
    // kernels.hpp:155-162: tmp = A x ; res = b - tmp
    bis_status st = bis_spmv(ctx, A, x, tmp);
    if (st != BIS_OK) return st;
    return bis_subtract_vectors(ctx, res, b, tmp, A->n_rows, 1.0);
}

} // extern "C"
