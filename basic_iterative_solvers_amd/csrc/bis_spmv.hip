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
// MODE 0: y = A x.  MODE 1: also partials[b] = sum y[r]*w[r] (CG's (Ap,p)).
// MODE 2: triangular-sweep epilogue for one dependency level given as a row
// range: y[r] = (w[r] - (A x)[r]) / dinv_or_d[r], i.e. x_level = (b - T x)/D
// (kernels.hpp:70,102); y may be the same array as x -- rows of one level do
// not reference each other.
template <typename RP, int T, int U, bool NT, int MODE>
__global__ __launch_bounds__(T) void spmv_rowblock_kernel(
    const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const double *__restrict__ val, const double *x, double *y,
    const int32_t *__restrict__ blk_row, const int64_t *__restrict__ blk_nnz, int n_blocks,
    int n_blocks_pad8, const double *w, double *partials) {
    constexpr bool FUSE_DOT = MODE == 1;
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
                va[u].x *= x[c[u].x];
                va[u].y *= x[c[u].y];
                vb[u].x *= x[c[u].z];
                vb[u].y *= x[c[u].w];
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
        if (MODE == 2) y[r] = (w[r] - acc) / partials[r];
        else y[r] = acc;
        if (FUSE_DOT) dot_acc = fma(acc, w[r], dot_acc);
    }
    if (FUSE_DOT) {
        __shared__ double red[T / 64];
        const double t = block_sum<T>(dot_acc, red);
        if (threadIdx.x == 0) partials[b] = t;
    }
}



// "row-major" variant: phase 1 only STAGES the block's val/col stream in LDS
// (coalesced 16-byte loads, no gather); phase 2 gives each lane one row and
// walks it left to right: acc = fma(val, x[col], acc) -- CRS order with fma,
// bit-identical to the scalar reference loop.  Across the lanes of a wave the
// j-th gather then reads x[col_j(row)] of 64 CONSECUTIVE rows: for stencil-like
// matrices those addresses are consecutive, i.e. one coalesced 512-byte load
// instead of a 64-way scatter through the texture-address path.
// L lanes share a row (L = 1, 2, 4, 8 by mean row length): each takes a
// contiguous 1/L of the row, the L partial sums are combined left to right.
template <typename RP, int T, int U, int MODE, int L>
__global__ __launch_bounds__(T) void spmv_rowmajor_kernel(
    const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const double *__restrict__ val, const double *x, double *y,
    const int32_t *__restrict__ blk_row, const int64_t *__restrict__ blk_nnz, int n_blocks,
    int n_blocks_pad8, const double *w, double *partials, int cap) {
    constexpr bool FUSE_DOT = MODE == 1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *lv = smem;                                   // [cap] values
    int *lc = reinterpret_cast<int *>(smem + cap);       // [cap] columns
    const int b = xcd_remap(blockIdx.x, n_blocks_pad8);
    if (b >= n_blocks) return;
    const int r0 = blk_row[b], r1 = blk_row[b + 1];
    const int64_t s = blk_nnz[b], e = blk_nnz[b + 1];
    const int64_t s4 = s & ~(int64_t)3;
    const int sub = (int)threadIdx.x % L;
    const int my_r = r0 + (int)threadIdx.x / L;
    RP rp_a = 0, rp_z = 0;
    if (my_r < r1) { rp_a = row_ptr[my_r]; rp_z = row_ptr[my_r + 1]; }
    for (int64_t k0 = s4 + 4 * (int64_t)threadIdx.x; k0 < e; k0 += 4 * T * U) {
        v4i c[U];
        v2d va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = k0 + (int64_t)u * 4 * T;
            if (k < e) {
                c[u] = *reinterpret_cast<const v4i *>(col + k);
                va[u] = *reinterpret_cast<const v2d *>(val + k);
                vb[u] = *reinterpret_cast<const v2d *>(val + k + 2);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = k0 + (int64_t)u * 4 * T;
            if (k < e) {
                *reinterpret_cast<v4i *>(lc + (k - s4)) = c[u];
                v2d *dst = reinterpret_cast<v2d *>(lv + (k - s4));
                dst[0] = va[u];
                dst[1] = vb[u];
            }
        }
    }
    __syncthreads();
    double dot_acc = 0.0;
    // (r1 - r0) rounded up so that all L lanes of a row group stay in the loop together
    for (int r = my_r; r < r1; r += T / L) { // the L lanes of a group share r
        if (r != my_r) { rp_a = row_ptr[r]; rp_z = row_ptr[r + 1]; }
        int a = (int)((int64_t)rp_a - s4), z = (int)((int64_t)rp_z - s4);
        if (L > 1) { // this lane's contiguous share of the row
            const int q = (z - a + L - 1) / L;
            a = min(a + sub * q, z);
            z = min(a + q, z);
        }
        double acc = 0.0;
        int j = a;
        for (; j + 4 <= z; j += 4) { // 4 gathers in flight, consumed in order
            const int c0 = lc[j], c1 = lc[j + 1], c2 = lc[j + 2], c3 = lc[j + 3];
            const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
            acc = fma(lv[j], x0, acc);
            acc = fma(lv[j + 1], x1, acc);
            acc = fma(lv[j + 2], x2, acc);
            acc = fma(lv[j + 3], x3, acc);
        }
        for (; j < z; ++j) acc = fma(lv[j], x[lc[j]], acc);
        if (L > 1) { // combine the L shares left to right on the group's first lane
            double tot = acc;
#pragma unroll
            for (int i = 1; i < L; ++i) tot += __shfl_down(acc, i, L);
            acc = tot;
        }
        if (sub == 0) {
            if (MODE == 2) y[r] = (w[r] - acc) / partials[r];
            else y[r] = acc;
            if (FUSE_DOT) dot_acc = fma(acc, w[r], dot_acc);
        }
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


// ---------------------------------------------------------------------------
// x-window variant.  Analysis (once per matrix): for every row block, the
// sorted list of the 16-column tiles of x its non-zeros reference (<= 128
// tiles) and, per non-zero, a 16-bit offset into that window.  The SpMV then
//   phase 0  copies the block's x tiles into LDS with coalesced loads (each x
//            entry once per block instead of one L1 gather per non-zero),
//   phase 1  streams val (8 B) + offset (2 B) -- 10 B per non-zero instead of
//            CRS's 12 -- and takes x from LDS,
//   phase 2  sums rows from LDS as before.
// Products, their order and the row sums are those of the CRS kernel; the CRS
// arrays stay authoritative (download, split, triangular solves use them).
// A matrix with a block touching more than 64 tiles keeps the gather kernel.
// ---------------------------------------------------------------------------
constexpr int kHash = 512;

template <typename RP>
__global__ __launch_bounds__(256) void window_build_kernel(
    const int32_t *__restrict__ col, const int64_t *__restrict__ blk_nnz, int n_blocks,
    int64_t loc_base, uint16_t *__restrict__ loc, int32_t *__restrict__ tiles,
    int32_t *__restrict__ tile_cnt, int *__restrict__ status /* [0]=overflow flag, [1]=max tiles */) {
    __shared__ int htab[kHash];
    __shared__ int list[kWinMaxTiles], sorted[kWinMaxTiles];
    __shared__ int cnt, overflow;
    const int b = blockIdx.x;
    if (b >= n_blocks) return;
    const int64_t s = blk_nnz[b], e = blk_nnz[b + 1];
    for (int i = threadIdx.x; i < kHash; i += 256) htab[i] = -1;
    if (threadIdx.x == 0) { cnt = 0; overflow = 0; }
    __syncthreads();
    for (int64_t k = s + threadIdx.x; k < e; k += 256) {
        const int tile = col[k] / kWinTile;
        unsigned h = ((unsigned)tile * 2654435761u) >> 23; // 9 bits -> kHash
        for (int probe = 0; probe < kHash; ++probe) {
            const int old = atomicCAS(&htab[h], -1, tile);
            if (old == tile) break;
            if (old == -1) {
                const int slot = atomicAdd(&cnt, 1);
                if (slot < kWinMaxTiles) list[slot] = tile; else overflow = 1;
                break;
            }
            h = (h + 1) & (kHash - 1);
            if (probe == kHash - 1) overflow = 1;
        }
    }
    __syncthreads();
    if (overflow) {
        if (threadIdx.x == 0) { tile_cnt[b] = -1; atomicExch(&status[0], 1); }
        return;
    }
    const int n = cnt;
    if ((int)threadIdx.x < n) { // rank sort of distinct values
        const int v = list[threadIdx.x];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += list[j] < v;
        sorted[rank] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < n) tiles[(size_t)b * kWinMaxTiles + threadIdx.x] = sorted[threadIdx.x];
    if (threadIdx.x == 0) { tile_cnt[b] = n; atomicMax(&status[1], n); }
    for (int64_t k = s + threadIdx.x; k < e; k += 256) {
        const int c = col[k], tile = c / kWinTile;
        int lo = 0, hi = n - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted[mid] < tile) lo = mid + 1; else hi = mid; }
        loc[k - loc_base] = (uint16_t)(lo * kWinTile + (c - tile * kWinTile));
    }
}

typedef unsigned short v4us __attribute__((ext_vector_type(4)));

template <typename RP, int T, int U, bool FUSE_DOT>
__global__ __launch_bounds__(T) void spmv_window_kernel(
    const RP *__restrict__ row_ptr, const uint16_t *__restrict__ loc, int64_t loc_base,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const int32_t *__restrict__ blk_row, const int64_t *__restrict__ blk_nnz,
    const int32_t *__restrict__ tiles, const int32_t *__restrict__ tile_cnt, int64_t n_cols,
    int xw_doubles, int n_blocks, int n_blocks_pad8, const double *__restrict__ w,
    double *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *xw = smem;                // [xw_doubles]  x window
    double *prod = smem + xw_doubles; // products
    const int b = xcd_remap(blockIdx.x, n_blocks_pad8);
    if (b >= n_blocks) return;
    const int r0 = blk_row[b], r1 = blk_row[b + 1];
    const int64_t s = blk_nnz[b], e = blk_nnz[b + 1];
    const int64_t s4 = s & ~(int64_t)3;
    const int nt = tile_cnt[b];
    const int my_r = r0 + (int)threadIdx.x;
    RP rp_a = 0, rp_z = 0;
    if (my_r < r1) { rp_a = row_ptr[my_r]; rp_z = row_ptr[my_r + 1]; }

    // issue the first stage of the val/offset stream before the window fill
    v4us lc[U];
    v2d va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t k = s4 + 4 * (int64_t)threadIdx.x + (int64_t)u * 4 * T;
        if (k < e) {
            lc[u] = *reinterpret_cast<const v4us *>(loc + (k - loc_base));
            va[u] = *reinterpret_cast<const v2d *>(val + k);
            vb[u] = *reinterpret_cast<const v2d *>(val + k + 2);
        }
    }
    // phase 0: x tiles -> LDS (coalesced 128 B per tile)
    for (int i = threadIdx.x; i < nt * kWinTile; i += T) {
        const int64_t c = (int64_t)tiles[(size_t)b * kWinMaxTiles + (i >> kWinTileLog)] * kWinTile + (i & (kWinTile - 1));
        xw[i] = c < n_cols ? x[c] : 0.0;
    }
    __syncthreads();
    // phase 1
    for (int64_t k0 = s4 + 4 * (int64_t)threadIdx.x; k0 < e; k0 += 4 * T * U) {
        if (k0 != s4 + 4 * (int64_t)threadIdx.x) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t k = k0 + (int64_t)u * 4 * T;
                if (k < e) {
                    lc[u] = *reinterpret_cast<const v4us *>(loc + (k - loc_base));
                    va[u] = *reinterpret_cast<const v2d *>(val + k);
                    vb[u] = *reinterpret_cast<const v2d *>(val + k + 2);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = k0 + (int64_t)u * 4 * T;
            if (k < e) {
                // elements before s / after e in the 4-aligned vector belong to a
                // neighbouring block (or the padding): their offsets are not for
                // this window -- clamp, the products are never read
                const int lim = nt * kWinTile - 1;
                va[u].x *= xw[min((int)lc[u].x, lim)];
                va[u].y *= xw[min((int)lc[u].y, lim)];
                vb[u].x *= xw[min((int)lc[u].z, lim)];
                vb[u].y *= xw[min((int)lc[u].w, lim)];
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

struct SpmvArgs {
    const void *row_ptr; const int32_t *col; const double *val; const double *x; double *y;
    const int32_t *blk_row; const int64_t *blk_nnz; int nb, nb8; const double *w; double *partials;
    size_t lds_bytes; hipStream_t stream; int mode; int lanes_per_row = 1;
};

template <typename RP, int T, int U, bool NT>
void launch_variant(const SpmvArgs &a) {
#define BIS_LV(MODE)                                                                              \
    hipLaunchKernelGGL((spmv_rowblock_kernel<RP, T, U, NT, MODE>), dim3(a.nb8), dim3(T), a.lds_bytes, \
                       a.stream, (const RP *)a.row_ptr, a.col, a.val, a.x, a.y, a.blk_row, a.blk_nnz, \
                       a.nb, a.nb8, a.w, a.partials)
    if (a.mode == 2) BIS_LV(2);
    else if (a.mode == 1) BIS_LV(1);
    else BIS_LV(0);
#undef BIS_LV
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
    case 60: case 61: { // row-major variant: LDS holds val (8 B) + col (4 B) per non-zero
        const int cap = (int)((a.lds_bytes / sizeof(double) + 1) & ~(size_t)1);
        const size_t lds = (size_t)cap * 12;
#define BIS_RM(MODE, LL)                                                                         \
    hipLaunchKernelGGL((spmv_rowmajor_kernel<RP, 256, 4, MODE, LL>), dim3(a.nb8), dim3(256), lds, \
                       a.stream, (const RP *)a.row_ptr, a.col, a.val, a.x, a.y, a.blk_row,        \
                       a.blk_nnz, a.nb, a.nb8, a.w, a.partials, cap)
#define BIS_RML(LL) do { if (a.mode == 2) BIS_RM(2, LL); else if (a.mode == 1) BIS_RM(1, LL); else BIS_RM(0, LL); } while (0)
        const int lanes = a.lanes_per_row;
        if (lanes >= 8) BIS_RML(8); else if (lanes >= 4) BIS_RML(4); else if (lanes >= 2) BIS_RML(2); else BIS_RML(1);
#undef BIS_RML
#undef BIS_RM
        return true;
    }
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

// lanes sharing a row in the row-major variant, by mean row length
int lanes_for(const bis_mat *A) {
    const double avg = A->n_rows ? (double)A->nnz / (double)A->n_rows : 0.0;
    return avg >= 48 ? 8 : avg >= 20 ? 4 : avg >= 10 ? 2 : 1;
}

// Default kernel by matrix shape (measured, gpurun A/B in one process):
//   short rows (mean < 10 nnz, e.g. the 7-point Anderson operator): the
//   row-major variant, 0.318 vs 0.338 ms on Anderson-256;
//   otherwise the product-staging variant (HPCG-256: 1.12 vs 1.23 ms).
int spmv_variant(const bis_mat *A) {
    if (bis_opts().spmv_variant >= 0) return bis_opts().spmv_variant;
    const double avg = A->n_rows ? (double)A->nnz / (double)A->n_rows : 0.0;
    return avg < 10.0 ? 60 : 40;
}

} // namespace

// The x-window variant is opt-in (spmv_window=1): measured 1.20-1.33 ms against
// 1.00-1.12 ms for the gather kernel on HPCG-256 (profiles/).
int spmv_window_mode() { return bis_opts().spmv_window < 0 ? 0 : bis_opts().spmv_window; }

bis_status bis_spmv_build_window(bis_ctx *ctx, bis_mat *A) {
    A->win_ok = false;
    if (!spmv_window_mode() || A->nnz == 0 || A->n_rows == 0) return BIS_OK;
    if ((int64_t)A->chunk_nnz + A->max_row_nnz + 8 > 6144) return BIS_OK; // LDS: products + 16 KiB window <= 64 KiB
    const int nb = A->n_blocks;
    int64_t ends[2];
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&ends[0], A->blk_nnz, 8, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&ends[1], A->blk_nnz + nb, 8, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    A->loc_base = ends[0] & ~(int64_t)3;
    const size_t n_loc = (size_t)(ends[1] - A->loc_base) + 16;
    hipFree(A->loc); hipFree(A->tiles); hipFree(A->tile_cnt);
    A->loc = nullptr; A->tiles = nullptr; A->tile_cnt = nullptr;
    BIS_HIP_CHECK(ctx, hipMalloc(&A->loc, sizeof(uint16_t) * n_loc));
    BIS_HIP_CHECK(ctx, hipMalloc(&A->tiles, sizeof(int32_t) * (size_t)nb * kWinMaxTiles));
    BIS_HIP_CHECK(ctx, hipMalloc(&A->tile_cnt, sizeof(int32_t) * (size_t)nb));
    BIS_HIP_CHECK(ctx, hipMemsetAsync(A->loc, 0, sizeof(uint16_t) * n_loc, ctx->stream));
    int *status = (int *)ctx->counters + 40;
    BIS_HIP_CHECK(ctx, hipMemsetAsync(status, 0, 2 * sizeof(int), ctx->stream));
    if (A->rp64)
        hipLaunchKernelGGL(window_build_kernel<int64_t>, dim3(nb), dim3(256), 0, ctx->stream, A->col,
                           A->blk_nnz, nb, A->loc_base, A->loc, A->tiles, A->tile_cnt, status);
    else
        hipLaunchKernelGGL(window_build_kernel<int32_t>, dim3(nb), dim3(256), 0, ctx->stream, A->col,
                           A->blk_nnz, nb, A->loc_base, A->loc, A->tiles, A->tile_cnt, status);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    int h[2] = {0, 0};
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(h, status, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h[0]) { // some block touches too many tiles: keep the gather kernel, drop the structure
        hipFree(A->loc); hipFree(A->tiles); hipFree(A->tile_cnt);
        A->loc = nullptr; A->tiles = nullptr; A->tile_cnt = nullptr;
        return BIS_OK;
    }
    A->max_tiles = h[1];
    A->win_ok = true;
    return BIS_OK;
}

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
    if (A->win_ok && spmv_window_mode()) {
        const int xw_doubles = ((A->max_tiles * kWinTile) + 1) & ~1;
        const size_t lds_win = lds_bytes + sizeof(double) * (size_t)xw_doubles;
        bis_prof_begin(ctx);
#define BIS_WIN_LAUNCH(RP, FUSE)                                                                   \
    hipLaunchKernelGGL((spmv_window_kernel<RP, 256, 2, FUSE>), dim3(nb8), dim3(256), lds_win,      \
                       ctx->stream, (const RP *)A->row_ptr, A->loc, A->loc_base, A->val, x, y,     \
                       A->blk_row, A->blk_nnz, A->tiles, A->tile_cnt, A->n_cols, xw_doubles, nb,   \
                       nb8, w, ctx->partials + partials_off)
        if (A->rp64) { if (w) BIS_WIN_LAUNCH(int64_t, true); else BIS_WIN_LAUNCH(int64_t, false); }
        else { if (w) BIS_WIN_LAUNCH(int32_t, true); else BIS_WIN_LAUNCH(int32_t, false); }
#undef BIS_WIN_LAUNCH
        bis_prof_end(ctx);
        BIS_HIP_CHECK(ctx, hipGetLastError());
        if (w && n_partials) *n_partials = nb;
        return BIS_OK;
    }
    SpmvArgs a{A->row_ptr, A->col, A->val, x, y, A->blk_row, A->blk_nnz, nb, nb8, w,
               ctx->partials + partials_off, lds_bytes,
               ctx->stream, w ? 1 : 0};
    a.lanes_per_row = lanes_for(A);
    bis_prof_begin(ctx);
    const bool ok = A->rp64 ? launch_by_id<int64_t>(spmv_variant(A), a)
                            : launch_by_id<int32_t>(spmv_variant(A), a);
    bis_prof_end(ctx);
    if (!ok) { ctx->err = "bis_spmv: unknown BIS_SPMV_VARIANT"; return BIS_ERR_INVALID; }
    BIS_HIP_CHECK(ctx, hipGetLastError());
    if (w && n_partials) *n_partials = nb;
    return BIS_OK;
}

// internal: one triangular-sweep level on the row range a view T covers:
// y[r] = (b[r] - (T x)[r]) / D[r]  (y, b, D already offset to the view's first row)
bis_status bis_spmv_trsv_level(bis_ctx *ctx, const bis_mat *T, const double *x, double *y,
                               const double *b, const double *D) {
    if (T->n_rows == 0) return BIS_OK;
    const int64_t lds_doubles = (int64_t)T->chunk_nnz + T->max_row_nnz + 8;
    const size_t lds_bytes = sizeof(double) * (size_t)lds_doubles;
    if (lds_bytes > 64 * 1024) { ctx->err = "sptrsv level: row too long for the streaming kernel"; return BIS_ERR_UNSUPPORTED; }
    const int nb = T->n_blocks, nb8 = (nb + 7) & ~7;
    SpmvArgs a{T->row_ptr, T->col, T->val, x, y, T->blk_row, T->blk_nnz, nb, nb8, b,
               const_cast<double *>(D), lds_bytes, ctx->stream, 2};
    a.lanes_per_row = lanes_for(T);
    const bool ok = T->rp64 ? launch_by_id<int64_t>(spmv_variant(T), a) : launch_by_id<int32_t>(spmv_variant(T), a);
    if (!ok) return BIS_ERR_INVALID;
    BIS_HIP_CHECK(ctx, hipGetLastError());
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
