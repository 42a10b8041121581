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

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool NT, typename V>
__device__ __forceinline__ V stream_load(const V *p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// T threads; U = 4-non-zero vectors staged per lane before the first use
// (all loads of a stage are in flight together); NT = nontemporal val/col
// loads (streamed once: keep them from evicting x out of L2).
template <typename RP, int T, int U, bool NT, bool FUSE_DOT, int DBG = 0>
__global__ __launch_bounds__(T) void spmv_rowblock_kernel(
    const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const int32_t *__restrict__ blk_row, const int64_t *__restrict__ blk_nnz, int n_blocks,
    int n_blocks_pad8, const double *__restrict__ w, double *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) double prod[];
    const int b = xcd_remap(blockIdx.x, n_blocks_pad8);
    if (b >= n_blocks) return;
    // one dependent level only: row range and nnz range come from the block
    // table; the row_ptr entries phase 2 needs are fetched now, under phase 1
    const int r0 = blk_row[b], r1 = blk_row[b + 1];
    const int64_t s = blk_nnz[b], e = blk_nnz[b + 1];
    const int64_t s4 = s & ~(int64_t)3;
    const int my_r = r0 + (int)threadIdx.x;
    RP rp_a = 0, rp_z = 0;
    if (my_r < r1) { rp_a = row_ptr[my_r]; rp_z = row_ptr[my_r + 1]; }

    // phase 1: stream val/col, gather x, park products
    for (int64_t k0 = s4 + 4 * (int64_t)threadIdx.x; k0 < e; k0 += 4 * T * U) {
        v4i c[U];
        v2d va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = k0 + (int64_t)u * 4 * T;
            if (k < e) {
                c[u] = stream_load<NT>(reinterpret_cast<const v4i *>(col + k));
                va[u] = stream_load<NT>(reinterpret_cast<const v2d *>(val + k));
                vb[u] = stream_load<NT>(reinterpret_cast<const v2d *>(val + k + 2));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = k0 + (int64_t)u * 4 * T;
            if (k < e) {
                if (DBG == 1) { // tuning probe: no gather at all
                    va[u].x *= (double)c[u].x; va[u].y *= (double)c[u].y;
                    vb[u].x *= (double)c[u].z; vb[u].y *= (double)c[u].w;
                } else if (DBG == 2) { // tuning probe: coalesced pseudo-gather
                    const v2d xa = *reinterpret_cast<const v2d *>(x + ((k + c[u].x * 0) & 0xFFFFF));
                    const v2d xb = *reinterpret_cast<const v2d *>(x + ((k + 2 + c[u].z * 0) & 0xFFFFF));
                    va[u] *= xa; vb[u] *= xb;
                } else {
                va[u].x *= x[c[u].x];
                va[u].y *= x[c[u].y];
                vb[u].x *= x[c[u].z];
                vb[u].y *= x[c[u].w];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = k0 + (int64_t)u * 4 * T;
            if (k < e) {
                v2d *dst = reinterpret_cast<v2d *>(prod + (k - s4));
                dst[0] = va[u];
                dst[1] = vb[u];
            }
        }
    }
    __syncthreads();

    // phase 2: one lane per row, left-to-right sum in CRS order
    double dot_acc = 0.0;
    for (int r = my_r; r < r1; r += T) {
        if (r != my_r) { rp_a = row_ptr[r]; rp_z = row_ptr[r + 1]; }
        const int a = (int)((int64_t)rp_a - s4), z = (int)((int64_t)rp_z - s4);
        double acc = 0.0;
        for (int j = a; j < z; ++j) acc += prod[j];
        y[r] = acc;
        if (FUSE_DOT) dot_acc = fma(acc, w[r], dot_acc);
    }
    if (FUSE_DOT) {
        __shared__ double red[T / 64];
        const double t = block_sum<T>(dot_acc, red);
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

struct SpmvArgs {
    const void *row_ptr; const int32_t *col; const double *val; const double *x; double *y;
    const int32_t *blk_row; const int64_t *blk_nnz; int nb, nb8; const double *w; double *partials;
    size_t lds_bytes; hipStream_t stream;
};

template <typename RP, int T, int U, bool NT>
void launch_variant(const SpmvArgs &a) {
    if (a.w)
        hipLaunchKernelGGL((spmv_rowblock_kernel<RP, T, U, NT, true>), dim3(a.nb8), dim3(T),
                           a.lds_bytes, a.stream, (const RP *)a.row_ptr, a.col, a.val, a.x, a.y,
                           a.blk_row, a.blk_nnz, a.nb, a.nb8, a.w, a.partials);
    else
        hipLaunchKernelGGL((spmv_rowblock_kernel<RP, T, U, NT, false>), dim3(a.nb8), dim3(T),
                           a.lds_bytes, a.stream, (const RP *)a.row_ptr, a.col, a.val, a.x, a.y,
                           a.blk_row, a.blk_nnz, a.nb, a.nb8, a.w, a.partials);
}

// variant id = T/256-1 (0,1,3) * 100 + U * 10 + NT      (tuning knob BIS_SPMV_VARIANT)
template <typename RP>
bool launch_by_id(int id, const SpmvArgs &a) {
    switch (id) {
    case 10: launch_variant<RP, 256, 1, false>(a); return true;
    case 11: launch_variant<RP, 256, 1, true>(a); return true;
    case 20: launch_variant<RP, 256, 2, false>(a); return true;
    case 21: launch_variant<RP, 256, 2, true>(a); return true;
    case 40: launch_variant<RP, 256, 4, false>(a); return true;
    case 41: launch_variant<RP, 256, 4, true>(a); return true;
    case 9001: hipLaunchKernelGGL((spmv_rowblock_kernel<RP, 256, 4, false, false, 1>), dim3(a.nb8), dim3(256),
                           a.lds_bytes, a.stream, (const RP *)a.row_ptr, a.col, a.val, a.x, a.y,
                           a.blk_row, a.blk_nnz, a.nb, a.nb8, a.w, a.partials); return true;
    case 9002: hipLaunchKernelGGL((spmv_rowblock_kernel<RP, 256, 4, false, false, 2>), dim3(a.nb8), dim3(256),
                           a.lds_bytes, a.stream, (const RP *)a.row_ptr, a.col, a.val, a.x, a.y,
                           a.blk_row, a.blk_nnz, a.nb, a.nb8, a.w, a.partials); return true;
    case 1040: launch_variant<RP, 128, 4, false>(a); return true;
    case 1020: launch_variant<RP, 128, 2, false>(a); return true;
    case 2040: launch_variant<RP, 64, 4, false>(a); return true;
    case 2080: launch_variant<RP, 64, 8, false>(a); return true;
    case 120: launch_variant<RP, 512, 2, false>(a); return true;
    case 121: launch_variant<RP, 512, 2, true>(a); return true;
    case 140: launch_variant<RP, 512, 4, false>(a); return true;
    case 141: launch_variant<RP, 512, 4, true>(a); return true;
    case 320: launch_variant<RP, 1024, 2, false>(a); return true;
    case 321: launch_variant<RP, 1024, 2, true>(a); return true;
    default: return false;
    }
}

int spmv_variant() {
    static int v = -1;
    if (v < 0) {
        v = 40;
        if (const char *e = getenv("BIS_SPMV_VARIANT")) v = atoi(e);
    }
    return v;
}

} // namespace

// internal: y = A x, optionally partials[b] = sum_{r in block b} y[r]*w[r]
// (n_partials returns the number of partials written; 0 if not fused).
bis_status bis_spmv_launch(bis_ctx *ctx, const bis_mat *A, const double *x, double *y,
                           const double *w, int *n_partials, size_t partials_off) {
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
    if (w && partials_off + (size_t)nb > ctx->partials_cap) {
        ctx->err = "bis_spmv: partials buffer too small (internal)";
        return BIS_ERR_INVALID;
    }
    SpmvArgs a{A->row_ptr, A->col, A->val, x, y, A->blk_row, A->blk_nnz, nb, nb8, w,
               ctx->partials + partials_off, lds_bytes,
               ctx->stream};
    bis_prof_begin(ctx);
    const bool ok = A->rp64 ? launch_by_id<int64_t>(spmv_variant(), a)
                            : launch_by_id<int32_t>(spmv_variant(), a);
    bis_prof_end(ctx);
    if (!ok) { ctx->err = "bis_spmv: unknown BIS_SPMV_VARIANT"; return BIS_ERR_INVALID; }
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
