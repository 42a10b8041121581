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
//   * XCD-aware block map, fine-grained: workgroups are dealt round-robin to
//     the 8 XCDs, so within every window of 64 consecutive row blocks XCD j is
//     given the 8 consecutive blocks [8j, 8j+8) (neighbouring x-lines share one
//     L2) while all XCDs still stream ONE advancing window of val/col.  Giving
//     each XCD its own contiguous SLAB of the matrix -- the textbook map --
//     was measured 9 % slower (HPCG-256, same arrays: 1.109 ms slabs, 1.032
//     plain blockIdx order, 1.012 groups of 8): eight far-apart HBM streams
//     cost more than the x re-fetches they save, which the 256 MiB Infinity
//     Cache absorbs.  Option "spmv_xcd_remap": 0 none, 1 slabs, G>1 groups.
//
//   * packed column stream (default): per row block <= 8 column windows, per
//     non-zero a 16-bit code (window:3 | offset:13) -- 10 instead of 12 streamed
//     bytes per non-zero; the CRS arrays stay authoritative.  Falls back to the
//     32-bit columns when a block needs more windows.
//
//   * value dictionary (default where it applies): a matrix with at most 256
//     distinct values -- or whose entries off the diagonal have at most 255 --
//     streams a 1-byte value code per non-zero against a table in LDS: 3 bytes
//     per non-zero, bit-identical y.  Two kernels: lane-per-row with the codes
//     staged through LDS (spmv_rowmajor_vd_kernel, its own packing over 256-row
//     blocks, 8 windows of 8192 or 32 of 2048 columns) and the consecutive form
//     on the row-block tables (spmv_rowblock_vd_kernel).  See "Value-dictionary
//     variant" below and DESIGN.md section 4.
//
// Rows longer than the LDS budget fall back to a wave-per-row kernel.
// An optional fused epilogue accumulates sum_r y[r]*w[r] (the (Ap,p) of
// cg.hpp:23) into per-wave partials so CG needs no separate dot pass; a third
// epilogue turns the kernel into one triangular-sweep step on a row range.
// Measurements, the PMC analysis of what bounds the kernel (L1->L2 request rate,
// 93 % of the streaming rate in fabric bytes) and the tuning record: DESIGN.md 4.
#include "bis_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

typedef unsigned short v4us __attribute__((ext_vector_type(4)));

// x[c]: with fewer than 2^29 columns the byte offset fits 32 bits (one shift,
// scalar base + 32-bit vector offset addressing); WIDE keeps 64-bit arithmetic.
template <bool WIDE>
__device__ __forceinline__ double x_at(const char *xb, int c) {
    if (WIDE) return *reinterpret_cast<const double *>(xb + (int64_t)c * 8);
    return *reinterpret_cast<const double *>(xb + (uint32_t)((uint32_t)c << 3));
}

constexpr int kPkSegs = 8, kPkOffBits = 13, kPkSpan = 1 << kPkOffBits;
constexpr int kPk3Segs = 32, kPk3OffBits = 11; // second format: more, narrower windows (reordered matrices)

template <int PK>
__device__ __forceinline__ int pk_decode(unsigned code, const int (&B)[8], int lane_base, int col_max) {
    int base;
    if (PK == 3) { // 32 windows of 2048 columns: window:5 | offset:11, lane j < 32 of every wave holds base j
        base = __builtin_amdgcn_ds_bpermute((int)((code >> (kPk3OffBits - 2)) & 124u), lane_base);
        return min(base + (int)(code & ((1u << kPk3OffBits) - 1)), col_max);
    }
    if (PK == 2) {
        base = __builtin_amdgcn_ds_bpermute((int)((code >> (kPkOffBits - 2)) & 28u), lane_base);
    } else {
        const bool b0 = code & (1u << kPkOffBits), b1 = code & (2u << kPkOffBits), b2 = code & (4u << kPkOffBits);
        const int a0 = b0 ? B[1] : B[0], a1 = b0 ? B[3] : B[2], a2 = b0 ? B[5] : B[4], a3 = b0 ? B[7] : B[6];
        const int c0 = b1 ? a1 : a0, c1 = b1 ? a3 : a2;
        base = b2 ? c1 : c0;
    }
    return min(base + (int)(code & (kPkSpan - 1)), col_max);
}

// T threads; U = 4-non-zero vectors staged per lane before the first use
// (all loads of a stage are in flight together); NT = nontemporal val/col
// loads (streamed once: keep them from evicting x out of L2).
// MODE 0: y = A x.  MODE 1: also partials[b] = sum y[r]*w[r] (CG's (Ap,p)).
// MODE 2: triangular-sweep epilogue for one dependency level given as a row
// range: y[r] = (w[r] - (A x)[r]) / dinv_or_d[r], i.e. x_level = (b - T x)/D
// (kernels.hpp:70,102); y may be the same array as x -- rows of one level do
// not reference each other.
//
// PK != 0: the column stream is the packed one (2 B per non-zero): code =
// segment:3 | offset:13, column = seg_base[8*b + segment] + offset.  PK 1 picks
// the base with a select tree over 8 registers, PK 2 with one cross-lane
// permute (lane j < 8 of every wave holds base j).  Codes of the neighbouring
// blocks that share the first/last 4-aligned vector decode against the wrong
// bases: clamped to a valid column, their products are never read.
template <typename RP, int T, int U, int PK, int MODE, bool WIDE = false, int BR = 0>
__global__ __launch_bounds__(T) void spmv_rowblock_kernel(
    const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const double *__restrict__ val, const double *x, double *y,
    const int32_t *__restrict__ blk_row, const int64_t *__restrict__ blk_nnz, int n_blocks,
    int n_blocks_pad8, const double *w, double *partials, const uint16_t *__restrict__ pk,
    int64_t pk_base, const int32_t *__restrict__ seg_base, int col_max, const int *stop, int acc_y) {
    constexpr bool FUSE_DOT = MODE == 1;
    if (stop && stop[1]) return; // the solver has stopped: this launch is a no-op
    extern __shared__ __attribute__((aligned(16))) double prod[];
    const int b = n_blocks_pad8 > 0 ? xcd_remap(blockIdx.x, n_blocks_pad8)
                                    : (n_blocks_pad8 < -1 ? xcd_group_remap(blockIdx.x, -n_blocks_pad8) : (int)blockIdx.x);
    if (b >= n_blocks) return;
    // one dependent level only: row range and nnz range come from the block
    // table; the row_ptr entries phase 2 needs are fetched now, under phase 1
    const int r0 = blk_row[b], r1 = blk_row[b + 1];
    const int64_t s = blk_nnz[b], e = blk_nnz[b + 1];
    const int64_t s4 = s & ~(int64_t)3;
    const int my_r = r0 + (int)threadIdx.x;
    RP rp_a = 0, rp_z = 0;
    if (my_r < r1) { rp_a = row_ptr[my_r]; rp_z = row_ptr[my_r + 1]; }
    int segb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int lane_base = 0;
    if (PK == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) segb[i] = seg_base[(size_t)b * 8 + i];
    }
    if (PK == 2) lane_base = seg_base[(size_t)b * 8 + (threadIdx.x & 7)];
    if (PK == 3) lane_base = seg_base[(size_t)b * kPk3Segs + (threadIdx.x & (kPk3Segs - 1))];
    const char *xb = reinterpret_cast<const char *>(x);

    // phase 1: stream val/col, gather x, park products.  Branch-free up to the
    // LDS store: lanes past the block's end re-read its last vector (clamped
    // index), so every load and gather of all U stages is in flight before
    // the first use and the wave stays whole (the lane permute needs that).
    if (BR == 2) {
        // consecutive form: lane t of the workgroup takes non-zeros base + j*T + t, so
        // one gather instruction covers 64 consecutive non-zeros (neighbouring lanes
        // mostly share an x cache line); narrow (8 B / 4 B / 2 B) but fully coalesced
        // stream loads, conflict-free LDS stores.  No alignment slack: prod[k - s4]
        // keeps the index convention of the other forms.
        constexpr int J = 4 * U;
        for (int64_t base = s; base < e; base += (int64_t)J * T) {
            int cc[J];
            double vv[J], xx[J];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int64_t k = min(base + (int64_t)j * T + threadIdx.x, e - 1);
                if (PK) cc[j] = pk[k - pk_base];
                else cc[j] = col[k];
                vv[j] = val[k];
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                if (PK) cc[j] = pk_decode<PK>((unsigned)cc[j], segb, lane_base, col_max);
                xx[j] = x_at<WIDE>(xb, cc[j]);
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int64_t k = base + (int64_t)j * T + threadIdx.x;
                double pr = vv[j] * xx[j];
                asm volatile("" : "+v"(pr));
                if (k < e) prod[k - s4] = pr;
            }
        }
    } else if (BR == 1) { // staged, predicated form (each stage waits for its own gathers)
        for (int64_t k0 = s4 + 4 * (int64_t)threadIdx.x; k0 < e; k0 += 4 * T * U) {
            v4i c[U];
            v4us pc[U];
            v2d va[U], vb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t k = k0 + (int64_t)u * 4 * T;
                if (k < e) {
                    if (PK) pc[u] = *reinterpret_cast<const v4us *>(pk + (k - pk_base));
                    else c[u] = *reinterpret_cast<const v4i *>(col + k);
                    va[u] = *reinterpret_cast<const v2d *>(val + k);
                    vb[u] = *reinterpret_cast<const v2d *>(val + k + 2);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t k = k0 + (int64_t)u * 4 * T;
                if (k < e) {
                    if (PK) {
                        c[u].x = pk_decode<PK>(pc[u].x, segb, lane_base, col_max);
                        c[u].y = pk_decode<PK>(pc[u].y, segb, lane_base, col_max);
                        c[u].z = pk_decode<PK>(pc[u].z, segb, lane_base, col_max);
                        c[u].w = pk_decode<PK>(pc[u].w, segb, lane_base, col_max);
                    }
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
    } else {
    const int64_t k_last = (e - 1) & ~(int64_t)3;
    for (int64_t base = s4; base < e; base += 4 * T * U) {
        v4i c[U];
        v4us pc[U];
        v2d va[U], vb[U];
        double xv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = min(base + 4 * (int64_t)threadIdx.x + (int64_t)u * 4 * T, k_last);
            if (PK) pc[u] = *reinterpret_cast<const v4us *>(pk + (k - pk_base));
            else c[u] = *reinterpret_cast<const v4i *>(col + k);
            va[u] = *reinterpret_cast<const v2d *>(val + k);
            vb[u] = *reinterpret_cast<const v2d *>(val + k + 2);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (PK) {
                c[u].x = pk_decode<PK>(pc[u].x, segb, lane_base, col_max);
                c[u].y = pk_decode<PK>(pc[u].y, segb, lane_base, col_max);
                c[u].z = pk_decode<PK>(pc[u].z, segb, lane_base, col_max);
                c[u].w = pk_decode<PK>(pc[u].w, segb, lane_base, col_max);
            }
            xv[u][0] = x_at<WIDE>(xb, c[u].x);
            xv[u][1] = x_at<WIDE>(xb, c[u].y);
            xv[u][2] = x_at<WIDE>(xb, c[u].z);
            xv[u][3] = x_at<WIDE>(xb, c[u].w);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = base + 4 * (int64_t)threadIdx.x + (int64_t)u * 4 * T;
            double p0 = va[u].x * xv[u][0], p1 = va[u].y * xv[u][1];
            double p2 = vb[u].x * xv[u][2], p3 = vb[u].y * xv[u][3];
            // pin the products here: otherwise the val loads sink into the
            // predicated store below and leave the load stage
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            if (k < e) {
                v2d *dst = reinterpret_cast<v2d *>(prod + (k - s4));
                dst[0] = v2d{p0, p1};
                dst[1] = v2d{p2, p3};
            }
        }
    }
    }
    __syncthreads();

    // phase 2: one lane per row, left-to-right sum in CRS order
    double dot_acc = 0.0;
    for (int r = my_r; r < r1; r += T) {
        if (r != my_r) { rp_a = row_ptr[r]; rp_z = row_ptr[r + 1]; }
        const int a = (int)((int64_t)rp_a - s4), z = (int)((int64_t)rp_z - s4);
        // (acc_y: this launch is a later column slab of the matrix -- bis_spmv_slab.hip -- and continues the row's sum where the
        // slab before it left it in y)
        double acc = (MODE != 2 && acc_y) ? y[r] : 0.0;
        for (int j = a; j < z; ++j) acc += prod[j];
        if (MODE == 2) y[r] = (w[r] - acc) / partials[r];
        else y[r] = acc;
        if (FUSE_DOT) dot_acc = fma(acc, w[r], dot_acc);
    }
    if (FUSE_DOT) { // one partial per wave (no workgroup barrier): kSpmvWaves per row block
        const double t = wave_sum(dot_acc);
        if ((threadIdx.x & 63) == 0) partials[(size_t)b * (T / 64) + (threadIdx.x >> 6)] = t;
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
// Value-dictionary variant.  A matrix with at most 256 distinct values (any
// constant-coefficient stencil: HPCG has two) keeps them in a 256-entry table;
// per non-zero the kernel streams a 2-byte column code and a 1-byte value code
// -- 3 instead of CRS's 12 bytes -- and takes the value from an LDS copy of the
// table.  The values are the CRS ones bit for bit, products, their order and
// the row sums are those of spmv_rowblock_kernel: same results to the last bit.
// The CRS arrays stay authoritative (download, split, triangular solves use
// them).  With a quarter of the bytes the kernel is no longer HBM-bound: what
// counts is the number of wave instructions and of L1 tag lookups per non-zero
// (profiles/r02_g_spmv_pmc_*), hence the form below.
// ---------------------------------------------------------------------------

// acc += prod[a] + ... + prod[z - 1], left to right (the reference's summation order), the LDS reads issued eight at a time
__device__ __forceinline__ void acc_row(const double *prod, int a, int z, double &acc) {
    int j = a;
    for (; j + 8 <= z; j += 8) {
        double p[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) p[q] = prod[j + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += p[q];
    }
    for (; j < z; ++j) acc += prod[j];
}

// Consecutive form: lane t of the workgroup takes non-zeros s + j*256 + t, so one gather instruction covers 64
// CONSECUTIVE non-zeros (2-3 rows of a stencil), the narrow stream loads (2 B + 1 B per lane) are fully coalesced, and
// the column window base comes from one cross-lane permute (lane j < 8 of every wave holds base j) instead of the
// select tree: 159 M vector-ALU wave instructions per HPCG-256 launch against 273 M for the 4-non-zeros-per-lane form
// with the select tree (0.62 against 0.70 ms; a persistent, software-pipelined variant of this form: 0.71 ms).
template <typename RP, int MODE>
__global__ __launch_bounds__(256) void spmv_rowblock_vd_kernel(
    const RP *__restrict__ row_ptr, const double *x, double *y, const int32_t *__restrict__ blk_row,
    const int64_t *__restrict__ blk_nnz, int n_blocks, int n_blocks_pad8, const double *w, double *partials,
    const uint16_t *__restrict__ pk, int64_t pk_base, const int32_t *__restrict__ seg_base, int col_max, const int *stop,
    const uint8_t *__restrict__ vcode, int64_t vd_base, const double *__restrict__ vdict) {
    constexpr int T = 256, J = 8;
    constexpr bool FUSE_DOT = MODE == 1;
    if (stop && stop[1]) return;
    extern __shared__ __attribute__((aligned(16))) double prod[];
    __shared__ double dict[256];
    const int b = n_blocks_pad8 > 0 ? xcd_remap(blockIdx.x, n_blocks_pad8)
                                    : (n_blocks_pad8 < -1 ? xcd_group_remap(blockIdx.x, -n_blocks_pad8) : (int)blockIdx.x);
    if (b >= n_blocks) return;
    dict[threadIdx.x] = vdict[threadIdx.x];
    const int r0 = blk_row[b], r1 = blk_row[b + 1];
    const int64_t s = blk_nnz[b], e = blk_nnz[b + 1];
    const int my_r = r0 + (int)threadIdx.x;
    RP rp_a = 0, rp_z = 0;
    if (my_r < r1) { rp_a = row_ptr[my_r]; rp_z = row_ptr[my_r + 1]; }
    const int lane_base = seg_base[(size_t)b * 8 + (threadIdx.x & 7)];
    const char *xb = reinterpret_cast<const char *>(x);
    const uint16_t *pkp = pk + (s - pk_base);
    const uint8_t *vcp = vcode + (s - vd_base);
    const int n = (int)(e - s);
    __syncthreads();
    for (int base = 0; base < n; base += J * T) {
        unsigned cc[J], vc[J];
        double xx[J], vv[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const unsigned idx = (unsigned)min(base + j * T + (int)threadIdx.x, n - 1); // 32-bit offsets: scalar base + vector offset addressing
            cc[j] = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(pkp) + (idx << 1));
            vc[j] = *(vcp + idx);
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int wbase = __builtin_amdgcn_ds_bpermute((int)((cc[j] >> (kPkOffBits - 2)) & 28u), lane_base);
            xx[j] = x_at<false>(xb, wbase + (int)(cc[j] & (kPkSpan - 1)));
            vv[j] = dict[vc[j]];
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int idx = base + j * T + (int)threadIdx.x;
            double pr = vv[j] * xx[j];
            asm volatile("" : "+v"(pr));
            if (idx < n) prod[idx] = pr;
        }
    }
    __syncthreads();
    double dot_acc = 0.0;
    for (int r = my_r; r < r1; r += T) {
        if (r != my_r) { rp_a = row_ptr[r]; rp_z = row_ptr[r + 1]; }
        const int a = (int)((int64_t)rp_a - s), z = (int)((int64_t)rp_z - s);
        double acc = 0.0;
        acc_row(prod, a, z, acc);
        if (MODE == 2) y[r] = (w[r] - acc) / partials[r];
        else y[r] = acc;
        if (FUSE_DOT) dot_acc = fma(acc, w[r], dot_acc);
    }
    if (FUSE_DOT) {
        const double t = wave_sum(dot_acc);
        if ((threadIdx.x & 63) == 0) partials[(size_t)b * (T / 64) + (threadIdx.x >> 6)] = t;
    }
}

// Lane-per-row form of the dictionary kernel (same three epilogues): a workgroup takes 256 consecutive
// rows, copies their column and value codes into LDS with 16- and 8-byte vector loads (3 bytes per non-zero), and
// lane t then walks row t in CRS order, eight non-zeros at a time: code from LDS, window base by the cross-lane
// permute, x gathered -- neighbouring lanes are neighbouring rows, so for a stencil the 64 gathers of one instruction
// fall into consecutive x entries -- value from the table, acc += value * x (two roundings, like the product/sum
// passes of the other forms: bit-identical y).  No product round trip through LDS, one gather instruction per 64
// non-zeros as the only vector-memory instruction in the inner loop.
constexpr int kRmRows = 256;
constexpr int kRmMaxRow = 40; // LDS: 256 rows * 40 codes * 3 B = 30 KiB

__device__ __forceinline__ int wave_max_i(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}

// DIAG: value code 255 stands for the row's own diagonal value vdiag[row] (matrices whose off-diagonal values are few
// but whose diagonal is not: Anderson's random potential).
// W32: the block's columns are coded against 32 windows of 2048 columns (lane j < 32 of every wave holds base j) instead
// of 8 windows of 8192: multi-colour-permuted matrices, whose rows reach into a dozen or more short column runs.
template <typename RP, int MODE, bool DIAG, bool W32>
__global__ __launch_bounds__(256) void spmv_rowmajor_vd_kernel(
    const RP *__restrict__ row_ptr, const double *x, double *y, int64_t n_rows, int n_blocks, int n_blocks_pad8,
    const double *w, double *partials, const uint16_t *__restrict__ pk, int64_t pk_base, const int32_t *__restrict__ seg_base,
    const int *stop, const uint8_t *__restrict__ vcode, int64_t vd_base, const double *__restrict__ vdict, int code_cap,
    const double *__restrict__ vdiag) {
    constexpr bool FUSE_DOT = MODE == 1;
    if (stop && stop[1]) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ double dict[256];
    const int b = n_blocks_pad8 > 0 ? xcd_remap(blockIdx.x, n_blocks_pad8)
                                    : (n_blocks_pad8 < -1 ? xcd_group_remap(blockIdx.x, -n_blocks_pad8) : (int)blockIdx.x);
    if (b >= n_blocks) return;
    dict[threadIdx.x] = vdict[threadIdx.x];
    const int64_t r0 = (int64_t)b * kRmRows;
    const int rows = (int)min((int64_t)kRmRows, n_rows - r0);
    const int64_t s = (int64_t)row_ptr[r0], e = (int64_t)row_ptr[r0 + rows];
    const int64_t s8 = s & ~(int64_t)7;
    const int first = (int)(s - s8);  // LDS position of the block's first code
    const int n8 = (int)(e - s8);     // LDS positions [0, n8) get loaded (rounded up to whole vectors of 8)
    uint16_t *lpk = reinterpret_cast<uint16_t *>(lds_raw);
    uint8_t *lvc = lds_raw + 2 * (size_t)code_cap;
    {
        const uint4 *gpk = reinterpret_cast<const uint4 *>(pk + (s8 - pk_base));
        const uint2 *gvc = reinterpret_cast<const uint2 *>(vcode + (s8 - vd_base));
        for (int v = (int)threadIdx.x; v * 8 < n8; v += 256) {
            reinterpret_cast<uint4 *>(lpk)[v] = gpk[v];
            reinterpret_cast<uint2 *>(lvc)[v] = gvc[v];
        }
    }
    const bool mine = (int)threadIdx.x < rows;
    int a = first, len = 0;
    double dval = 0.0;
    if (mine) {
        const int64_t ra = (int64_t)row_ptr[r0 + threadIdx.x];
        a = (int)(ra - s8);
        len = (int)((int64_t)row_ptr[r0 + threadIdx.x + 1] - ra);
        if (DIAG) dval = vdiag[r0 + threadIdx.x];
    }
    constexpr int SEGS = W32 ? kPk3Segs : kPkSegs, OFFBITS = W32 ? kPk3OffBits : kPkOffBits;
    const int lane_base = seg_base[(size_t)b * SEGS + (threadIdx.x & (SEGS - 1))];
    const char *xb = reinterpret_cast<const char *>(x);
    __syncthreads();
    double acc = 0.0;
    const int wmax = __builtin_amdgcn_readfirstlane(wave_max_i(len)), wmin = __builtin_amdgcn_readfirstlane(wave_min_i(len)); // wave-uniform
    // eight entries of every row at a time: codes from LDS, window base, gather, table value ...
    auto issue = [&](int j0, double (&xx)[8], double (&vv)[8]) {
        const bool whole = j0 + 8 <= wmin; // every row of the wave has these eight entries
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // ragged tail: lanes past their row's end re-read the block's first code; their product is dropped
            const int idx = (whole || j0 + q < len) ? a + j0 + q : first;
            const unsigned code = lpk[idx];
            const int wbase = __builtin_amdgcn_ds_bpermute((int)((code >> (OFFBITS - 2)) & (unsigned)((SEGS - 1) * 4)), lane_base);
            xx[q] = x_at<false>(xb, wbase + (int)(code & ((1u << OFFBITS) - 1)));
            const unsigned vcd = lvc[idx];
            vv[q] = dict[vcd];
            if (DIAG) vv[q] = vcd == 255u ? dval : vv[q];
        }
    };
    // ... and acc += value * x in CRS order
    auto consume = [&](int j0, const double (&xx)[8], const double (&vv)[8]) {
        if (j0 + 8 <= wmin) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double pr = vv[q] * xx[q];
                asm volatile("" : "+v"(pr));
                acc += pr;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double pr = vv[q] * xx[q];
                asm volatile("" : "+v"(pr));
                if (j0 + q < len) acc += pr;
            }
        }
    };
    // (issuing the next eight gathers before consuming the current eight -- sixteen in flight per lane -- was measured
    // slower: HPCG-256 0.513 against 0.476 ms, Anderson-256 0.246 against 0.197: the kernel is bound by the gather and
    // LDS instruction rates, not by latency)
    for (int j0 = 0; j0 < wmax; j0 += 8) {
        double xx[8], vv[8];
        issue(j0, xx, vv);
        consume(j0, xx, vv);
    }
    if (mine) y[r0 + threadIdx.x] = MODE == 2 ? (w[r0 + threadIdx.x] - acc) / partials[r0 + threadIdx.x] : acc; // MODE 2: one triangular-sweep level
    if (FUSE_DOT) {
        const double t = wave_sum(mine ? acc * w[r0 + threadIdx.x] : 0.0);
        if ((threadIdx.x & 63) == 0) partials[(size_t)b * 4 + (threadIdx.x >> 6)] = t;
    }
}

// rm_nnz[k] = row_ptr[min(256 k, n_rows)]
template <typename RP>
__global__ __launch_bounds__(256) void rm_blocks_kernel(const RP *__restrict__ row_ptr, int64_t n_rows, int n_blocks, int64_t *__restrict__ rm_nnz) {
    const int k = (int)(blockIdx.x * 256 + threadIdx.x);
    if (k <= n_blocks) rm_nnz[k] = (int64_t)row_ptr[min((int64_t)k * kRmRows, n_rows)];
}

// distinct values of val[s, e), one list per wave (a workgroup is one wave): lists[w * 257] = count (257 = more than
// 256, *overflow is raised and every wave stops), then the values' bit patterns
__global__ __launch_bounds__(64) void vd_collect_kernel(const double *__restrict__ val, int64_t s, int64_t e,
                                                        unsigned long long *__restrict__ lists, int *overflow) {
    __shared__ unsigned long long list[256];
    const int lane = threadIdx.x;
    int n = 0;
    for (int64_t k0 = s + (int64_t)blockIdx.x * 64; k0 < e && n <= 256; k0 += (int64_t)gridDim.x * 64) {
        if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { n = 257; break; }
        const int64_t k = k0 + lane;
        const unsigned long long v = k < e ? (unsigned long long)__double_as_longlong(val[k]) : 0ull;
        bool found = k >= e;
        for (int j = 0; j < n && __ballot(!found); ++j) found |= list[j] == v;
        while (const unsigned long long open = __ballot(!found)) {
            const int leader = (int)__builtin_ctzll(open);
            const unsigned long long lv = __shfl(v, leader);
            if (n == 256) { n = 257; break; }
            if (lane == 0) list[n] = lv;
            ++n;
            found |= v == lv;
        }
        __syncthreads(); // one wave: orders the list writes before the next round's reads
    }
    if (n > 256 && lane == 0) atomicExch(overflow, 1);
    if (lane == 0) lists[(size_t)blockIdx.x * 257] = (unsigned long long)n;
    for (int j = lane; j < n && j < 256; j += 64) lists[(size_t)blockIdx.x * 257 + 1 + j] = list[j];
}

// Row-wise variants for the "dictionary + per-row diagonal" encoding: a lane per row, diagonal entries (col == row +
// row0) are left out of the dictionary.  *overflow: 1 = more than 255 distinct off-diagonal values, 2 = a row with more
// than one diagonal entry (its values could differ: not representable).
template <typename RP>
__global__ __launch_bounds__(64) void vd_collect_rows_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                             const double *__restrict__ val, int64_t n_rows, int64_t row0,
                                                             unsigned long long *__restrict__ lists, int *overflow) {
    __shared__ unsigned long long list[256];
    const int lane = threadIdx.x;
    int n = 0;
    for (int64_t rb = (int64_t)blockIdx.x * 64; rb < n_rows && n <= 255; rb += (int64_t)gridDim.x * 64) {
        if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { n = 256; break; }
        const int64_t r = rb + lane;
        int64_t k = 0, k1 = 0;
        if (r < n_rows) { k = (int64_t)row_ptr[r]; k1 = (int64_t)row_ptr[r + 1]; }
        int n_diag = 0;
        while (__ballot(k < k1) && n <= 255) {
            bool found = true;
            unsigned long long v = 0;
            if (k < k1) {
                if ((int64_t)col[k] == r + row0) ++n_diag;
                else { v = (unsigned long long)__double_as_longlong(val[k]); found = false; }
                ++k;
            }
            for (int j = 0; j < n && __ballot(!found); ++j) found |= list[j] == v;
            while (const unsigned long long open = __ballot(!found)) {
                const int leader = (int)__builtin_ctzll(open);
                const unsigned long long lv = __shfl(v, leader);
                if (n == 255) { n = 256; break; }
                if (lane == 0) list[n] = lv;
                ++n;
                found |= v == lv;
            }
            __syncthreads();
        }
        if (__ballot(n_diag > 1)) { if (lane == 0) atomicExch(overflow, 2); n = 256; }
    }
    if (n > 255 && lane == 0) atomicCAS(overflow, 0, 1);
    if (lane == 0) lists[(size_t)blockIdx.x * 257] = (unsigned long long)n;
    for (int j = lane; j < n && j < 255; j += 64) lists[(size_t)blockIdx.x * 257 + 1 + j] = list[j];
}

template <typename RP>
__global__ __launch_bounds__(256) void vd_encode_rows_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                             const double *__restrict__ val, int64_t n_rows, int64_t row0, int64_t base,
                                                             const double *__restrict__ vdict, int n_dict, uint8_t *__restrict__ vcode,
                                                             double *__restrict__ vdiag) {
    __shared__ unsigned long long dict[256];
    dict[threadIdx.x] = (unsigned long long)__double_as_longlong(vdict[threadIdx.x]);
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += stride) {
        double d = 0.0;
        for (int64_t k = (int64_t)row_ptr[r]; k < (int64_t)row_ptr[r + 1]; ++k) {
            if ((int64_t)col[k] == r + row0) { d = val[k]; vcode[k - base] = 255; continue; }
            const unsigned long long v = (unsigned long long)__double_as_longlong(val[k]);
            int lo = 0, hi = n_dict - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (dict[mid] < v) lo = mid + 1; else hi = mid;
            }
            vcode[k - base] = (uint8_t)lo;
        }
        vdiag[r] = d;
    }
}

// vcode[k - base] = index of val[k] in the (ascending) dictionary; four codes per thread, one 32-bit store
__global__ __launch_bounds__(256) void vd_encode_kernel(const double *__restrict__ val, int64_t s, int64_t e, int64_t base,
                                                        const double *__restrict__ vdict, int n_dict, uint8_t *__restrict__ vcode) {
    __shared__ unsigned long long dict[256];
    dict[threadIdx.x] = (unsigned long long)__double_as_longlong(vdict[threadIdx.x]);
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t k4 = base + ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; k4 < e; k4 += stride) {
        unsigned packed = 0;
        for (int q = 0; q < 4; ++q) {
            const int64_t k = k4 + q;
            if (k < s || k >= e) continue;
            const unsigned long long v = (unsigned long long)__double_as_longlong(val[k]);
            int lo = 0, hi = n_dict - 1; // the value is in the table
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (dict[mid] < v) lo = mid + 1; else hi = mid;
            }
            packed |= (unsigned)lo << (8 * q);
        }
        *reinterpret_cast<unsigned *>(vcode + (k4 - base)) = packed;
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
    const int b = n_blocks_pad8 > 0 ? xcd_remap(blockIdx.x, n_blocks_pad8) : (int)blockIdx.x;
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
    if (FUSE_DOT) { // one partial per wave (no workgroup barrier): kSpmvWaves per row block
        const double t = wave_sum(dot_acc);
        if ((threadIdx.x & 63) == 0) partials[(size_t)b * (T / 64) + (threadIdx.x >> 6)] = t;
    }
}

// ---------------------------------------------------------------------------
// Packed-column analysis, one workgroup per row block: cover the block's
// columns greedily with windows [base, base + 8192) taken at the smallest
// uncovered column (so the bases come out ascending); more than 8 windows ->
// the matrix keeps the 32-bit column stream.
// ---------------------------------------------------------------------------
template <int SEGS, int OFFBITS>
__global__ __launch_bounds__(256) void pk_build_kernel(const int32_t *__restrict__ col,
                                                       const int64_t *__restrict__ blk_nnz, int n_blocks,
                                                       int64_t pk_base, uint16_t *__restrict__ pk,
                                                       int32_t *__restrict__ seg_base, int *__restrict__ status) {
    constexpr int SPAN = 1 << OFFBITS;
    __shared__ int base[SEGS];
    __shared__ int cur;
    const int b = blockIdx.x;
    if (b >= n_blocks) return;
    const int64_t s = blk_nnz[b], e = blk_nnz[b + 1];
    int n_seg = 0;
    for (int sg = 0; sg <= SEGS; ++sg) {
        if (threadIdx.x == 0) cur = INT32_MAX;
        __syncthreads();
        int m = INT32_MAX;
        for (int64_t k = s + threadIdx.x; k < e; k += 256) {
            const int c = col[k];
            bool covered = false;
            for (int i = 0; i < sg; ++i) covered |= (c >= base[i] && c - base[i] < SPAN);
            if (!covered) m = min(m, c);
        }
        if (m != INT32_MAX) atomicMin(&cur, m);
        __syncthreads();
        const int found = cur;
        __syncthreads();
        if (found == INT32_MAX) break;
        if (sg == SEGS) { // one window too many
            if (threadIdx.x == 0) atomicExch(&status[0], 1);
            return;
        }
        if (threadIdx.x == 0) base[sg] = found;
        n_seg = sg + 1;
        __syncthreads();
    }
    if (threadIdx.x < SEGS) seg_base[(size_t)b * SEGS + threadIdx.x] = (int)threadIdx.x < n_seg ? base[threadIdx.x] : 0;
    for (int64_t k = s + threadIdx.x; k < e; k += 256) {
        const int c = col[k];
        int sg = 0;
        for (int i = 1; i < n_seg; ++i) sg += base[i] <= c; // ascending bases: last one not above c
        pk[k - pk_base] = (uint16_t)((sg << OFFBITS) | (c - base[sg]));
    }
}

// A cheap look before pk_build_kernel: a window of SPAN columns overlaps at most two SPAN-aligned buckets, so a block whose
// entries fall into more than 2 * SEGS distinct buckets cannot be covered by SEGS windows.  64 blocks spread over the matrix
// are looked at (one pass each, a bitmap in LDS); if one of them proves the failure, the build -- up to SEGS + 1 passes over
// EVERY block before it gives up -- is skipped: a matrix without locality (`unstr:80,80,80` as generated, its triangles, its
// column slabs: 23 failing builds, 16.5 ms) is told so at once.  The build's own answer is never changed, only anticipated.
template <int SEGS, int OFFBITS>
__global__ __launch_bounds__(256) void pk_probe_kernel(const int32_t *__restrict__ col, const int64_t *__restrict__ blk_nnz, int n_blocks,
                                                       int n_words, int *__restrict__ status) {
    extern __shared__ unsigned bitmap[];
    __shared__ int s_count;
    const int b = (int)(((int64_t)blockIdx.x * n_blocks) / gridDim.x);
    for (int i = threadIdx.x; i < n_words; i += 256) bitmap[i] = 0u;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const int64_t s = blk_nnz[b], e = blk_nnz[b + 1];
    for (int64_t k = s + threadIdx.x; k < e; k += 256) {
        const unsigned bucket = min((unsigned)col[k] >> OFFBITS, (unsigned)n_words * 32u - 1u);
        atomicOr(&bitmap[bucket >> 5], 1u << (bucket & 31u));
    }
    __syncthreads();
    int c = 0;
    for (int i = threadIdx.x; i < n_words; i += 256) c += __popc(bitmap[i]);
    if (c) atomicAdd(&s_count, c);
    __syncthreads();
    if (threadIdx.x == 0 && s_count > 2 * SEGS) atomicExch(status, 1);
}

struct SpmvArgs {
    const void *row_ptr; const int32_t *col; const double *val; const double *x; double *y;
    const int32_t *blk_row; const int64_t *blk_nnz; int nb, nb8; const double *w; double *partials;
    size_t lds_bytes; hipStream_t stream; int mode; int n_cus = 256; bool remap = false; int remap_arg = -1; int grid = 0;
    const uint16_t *pk = nullptr; int64_t pk_base = 0; const int32_t *seg_base = nullptr; int col_max = 0; int pk_mode = 0;
    bool wide = false; // 2^29 columns or more: 64-bit x addressing, 32-bit column stream
    const int *stop = nullptr;
    const uint8_t *vcode = nullptr; int64_t vd_base = 0; const double *vdict = nullptr; // value dictionary (pk_mode 1 only)
    bool vd_rm_only = false; // lane-per-row kernel only
    int acc_y = 0;           // row sums start from y (a later column slab)
};

template <typename RP, int T, int U, int BR = 0>
void launch_variant(const SpmvArgs &a) {
#define BIS_LV(PK, MODE)                                                                          \
    hipLaunchKernelGGL((spmv_rowblock_kernel<RP, T, U, (PK) < 0 ? 0 : (PK), MODE, (PK) < 0, BR>), dim3(a.grid), dim3(T), a.lds_bytes, \
                       a.stream, (const RP *)a.row_ptr, a.col, a.val, a.x, a.y, a.blk_row, a.blk_nnz, \
                       a.nb, a.remap_arg, a.w, a.partials, a.pk, a.pk_base, a.seg_base, a.col_max, a.stop, a.acc_y)
#define BIS_LVM(PK)                                                                               \
    do {                                                                                          \
        if (a.mode == 2) BIS_LV(PK, 2);                                                           \
        else if (a.mode == 1) BIS_LV(PK, 1);                                                      \
        else BIS_LV(PK, 0);                                                                       \
    } while (0)
    if (a.wide) BIS_LVM(-1);
    else if (a.pk_mode == 1) BIS_LVM(1);
    else if (a.pk_mode == 2) BIS_LVM(2);
    else if (a.pk_mode == 3) BIS_LVM(3);
    else BIS_LVM(0);
#undef BIS_LVM
#undef BIS_LV
}

// variant id: 10/20/40 = 256 threads with U = 1/2/4 staged vectors per lane
// (tuning knob BIS_SPMV_VARIANT)
template <typename RP, int MODE>
void launch_vd(const SpmvArgs &a) {
    hipLaunchKernelGGL((spmv_rowblock_vd_kernel<RP, MODE>), dim3(a.grid), dim3(256), a.lds_bytes, a.stream,
                       (const RP *)a.row_ptr, a.x, a.y, a.blk_row, a.blk_nnz, a.nb, a.remap_arg, a.w, a.partials, a.pk,
                       a.pk_base, a.seg_base, a.col_max, a.stop, a.vcode, a.vd_base, a.vdict);
}

template <typename RP>
bool launch_by_id(int id, const SpmvArgs &a) {
    if (a.vcode && !a.acc_y && !a.vd_rm_only && id == 20 && a.pk_mode == 1) { // value dictionary, consecutive form: the default variant only
        if (a.mode == 2) launch_vd<RP, 2>(a);
        else if (a.mode == 1) launch_vd<RP, 1>(a);
        else launch_vd<RP, 0>(a);
        return true;
    }
    switch (id) {
    case 10: launch_variant<RP, 256, 1>(a); return true;
    case 20: launch_variant<RP, 256, 2>(a); return true;
    case 40: launch_variant<RP, 256, 4>(a); return true;
    case 11: launch_variant<RP, 256, 1, 1>(a); return true;
    case 21: launch_variant<RP, 256, 2, 1>(a); return true;
    case 41: launch_variant<RP, 256, 4, 1>(a); return true;
    case 12: launch_variant<RP, 256, 1, 2>(a); return true;
    case 22: launch_variant<RP, 256, 2, 2>(a); return true;
    default: return false;
    }
}

// Default: 256 threads, up to 4 staged vectors per lane (variant 40).  Tuning
// history (tools/spmv_ab.py, same arrays, interleaved rounds, HPCG-256):
// chunk 1024 1.019 ms | 1280 1.026 | 1536 1.047 | 2048 1.084 | 4096 1.100;
// 128-thread blocks 1.10; a lane-per-row "row-major" phase 2 (fma order) 1.23-1.30;
// a persistent, software-pipelined grid 1.23-1.48; 64 consecutive non-zeros per
// gather instruction (8/4-byte loads) 1.23; nontemporal val/col loads +10 %.
// kernel argument encoding of the block map: nb8 = XCD slabs, -1 = blockIdx order, -G = groups of G
// default: groups of 8 (HPCG-256: 1.012 ms vs 1.032 blockIdx order vs 1.109 slabs)
int remap_arg_for(int nb8) {
    const int r = bis_opts().spmv_xcd_remap < 0 ? 8 : bis_opts().spmv_xcd_remap;
    return r == 1 ? nb8 : (r > 1 ? -r : -1);
}

// grid size: a whole number of windows so that every logical block id < n_blocks is produced
int grid_for_map(int nb, int remap_arg) {
    const int w = remap_arg < -1 ? 8 * -remap_arg : 8;
    return (nb + w - 1) / w * w;
}

// threads per workgroup of a variant id (the fused dot writes one partial per wave)
int fused_threads(int) { return 256; }

// default: the branch-free 2-stage form with the packed stream, the staged 4-deep form with 32-bit columns
int spmv_variant(const SpmvArgs &a) {
    const int v = bis_opts().spmv_variant >= 0 ? bis_opts().spmv_variant : (a.pk_mode ? 20 : 41);
    return (a.pk_mode >= 2 && v % 10 != 0) ? 20 : v; // the lane-permute decodes need the whole wave: branch-free form only
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

void bis_spmv_drop_packed(bis_mat *A) {
    for (int t = 0; t < 2; ++t) {
        hipFree(A->pk[t]); hipFree(A->pk_seg[t]);
        A->pk[t] = nullptr; A->pk_seg[t] = nullptr; A->pk_state[t] = 0;
    }
}

void bis_spmv_drop_valdict(bis_mat *A) {
    hipFree(A->vcode); hipFree(A->vdict); hipFree(A->vdiag);
    A->vcode = nullptr; A->vdict = nullptr; A->vdiag = nullptr; A->vd_diag = false; A->vd_state = 0; A->vd_n = 0;
    hipFree(A->rm_nnz); hipFree(A->rm_pk); hipFree(A->rm_seg);
    A->rm_nnz = nullptr; A->rm_pk = nullptr; A->rm_seg = nullptr; A->rm_state = 0; A->rm_blocks = 0;
    bis_spmv_sellwin_drop(A);
    bis_spmv_win8_drop(A);
    bis_spmv_colslab_drop(A);
}

// the 256-row blocks of the lane-per-row form and their packed column stream; rm_state tells the outcome
static bis_status spmv_try_rowmajor(bis_ctx *ctx, bis_mat *A) {
    if (A->rm_state != 0) return BIS_OK;
    A->rm_state = -1;
    if (A->vd_state != 1 || A->n_rows == 0 || A->max_row_nnz > kRmMaxRow || A->n_cols >= ((int64_t)1 << 29)) return BIS_OK;
    const int64_t nb64 = (A->n_rows + kRmRows - 1) / kRmRows;
    if (nb64 > (int64_t)1 << 28) return BIS_OK;
    const int nb = (int)nb64;
    BIS_HIP_CHECK(ctx, hipMalloc(&A->rm_nnz, sizeof(int64_t) * (size_t)(nb + 1)));
    if (A->rp64) hipLaunchKernelGGL(rm_blocks_kernel<int64_t>, dim3((unsigned)(nb / 256 + 1)), dim3(256), 0, ctx->stream, (const int64_t *)A->row_ptr, A->n_rows, nb, A->rm_nnz);
    else hipLaunchKernelGGL(rm_blocks_kernel<int32_t>, dim3((unsigned)(nb / 256 + 1)), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr, A->n_rows, nb, A->rm_nnz);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    int64_t ends[2];
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&ends[0], A->rm_nnz, 8, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&ends[1], A->rm_nnz + nb, 8, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    A->rm_base = ends[0] & ~(int64_t)7;
    if (A->rm_base < A->vd_base) return BIS_OK; // (both are the first row's start rounded down: cannot happen)
    const size_t n_pk = (size_t)(ends[1] - A->rm_base) + 16;
    BIS_HIP_CHECK(ctx, hipMalloc(&A->rm_pk, sizeof(uint16_t) * n_pk));
    BIS_HIP_CHECK(ctx, hipMalloc(&A->rm_seg, sizeof(int32_t) * (size_t)nb * kPk3Segs));
    int *status = (int *)ctx->counters + 47;
    int h = 1;
    for (int kind = 1; kind <= 3 && h; kind += 2) {
        BIS_HIP_CHECK(ctx, hipMemsetAsync(A->rm_pk, 0, sizeof(uint16_t) * n_pk, ctx->stream));
        BIS_HIP_CHECK(ctx, hipMemsetAsync(status, 0, sizeof(int), ctx->stream));
        if (kind == 1)
            hipLaunchKernelGGL((pk_build_kernel<kPkSegs, kPkOffBits>), dim3(nb), dim3(256), 0, ctx->stream, A->col, A->rm_nnz, nb, A->rm_base,
                               A->rm_pk, A->rm_seg, status);
        else
            hipLaunchKernelGGL((pk_build_kernel<kPk3Segs, kPk3OffBits>), dim3(nb), dim3(256), 0, ctx->stream, A->col, A->rm_nnz, nb, A->rm_base,
                               A->rm_pk, A->rm_seg, status);
        BIS_HIP_CHECK(ctx, hipGetLastError());
        BIS_HIP_CHECK(ctx, hipMemcpyAsync(&h, status, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (!h) A->rm_kind = kind;
    }
    if (h) { // some block of 256 rows needs more column windows than either format has: the other kernels stay
        hipFree(A->rm_nnz); hipFree(A->rm_pk); hipFree(A->rm_seg);
        A->rm_nnz = nullptr; A->rm_pk = nullptr; A->rm_seg = nullptr;
        return BIS_OK;
    }
    A->rm_blocks = nb;
    A->rm_state = 1;
    return BIS_OK;
}

// 0 off, 1 consecutive form, 2 lane-per-row form for y = A x and the fused dot where the matrix qualifies
static int spmv_valdict_mode() { return bis_opts().spmv_valdict < 0 ? 2 : bis_opts().spmv_valdict; }

// distinct values of the matrix (plain), or of its off-diagonal entries (diag): at most `cap` bit patterns, ascending;
// *ok = false when there are more (or, diag, when a row has several diagonal entries)
static bis_status vd_collect(bis_ctx *ctx, const bis_mat *A, int64_t s, int64_t e, bool diag, std::vector<unsigned long long> *all, bool *ok) {
    *ok = false;
    all->clear();
    const size_t cap = diag ? 255 : 256;
    const int64_t units = diag ? A->n_rows : e - s;
    const int n_waves = (int)std::min<int64_t>((units + 63) / 64, (int64_t)ctx->n_cus * 8);
    unsigned long long *lists = nullptr;
    int *overflow = (int *)ctx->counters + 46;
    BIS_HIP_CHECK(ctx, hipMalloc(&lists, sizeof(unsigned long long) * 257 * (size_t)n_waves));
    hipError_t he = hipMemsetAsync(overflow, 0, sizeof(int), ctx->stream);
    if (he == hipSuccess) {
        if (!diag) hipLaunchKernelGGL(vd_collect_kernel, dim3(n_waves), dim3(64), 0, ctx->stream, A->val, s, e, lists, overflow);
        else if (A->rp64) hipLaunchKernelGGL(vd_collect_rows_kernel<int64_t>, dim3(n_waves), dim3(64), 0, ctx->stream, (const int64_t *)A->row_ptr, A->col, A->val, A->n_rows, A->view_row0, lists, overflow);
        else hipLaunchKernelGGL(vd_collect_rows_kernel<int32_t>, dim3(n_waves), dim3(64), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col, A->val, A->n_rows, A->view_row0, lists, overflow);
        he = hipGetLastError();
    }
    std::vector<unsigned long long> h((size_t)257 * n_waves);
    int h_over = 0;
    if (he == hipSuccess) he = hipMemcpyAsync(&h_over, overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(h.data(), lists, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
    hipFree(lists);
    BIS_HIP_CHECK(ctx, he);
    if (h_over) return BIS_OK;
    for (int w = 0; w < n_waves; ++w) {
        const size_t n = (size_t)h[(size_t)w * 257];
        if (n > cap) return BIS_OK;
        all->insert(all->end(), h.begin() + (size_t)w * 257 + 1, h.begin() + (size_t)w * 257 + 1 + n);
    }
    std::sort(all->begin(), all->end());
    all->erase(std::unique(all->begin(), all->end()), all->end());
    *ok = all->size() <= cap;
    return BIS_OK;
}

static bis_status spmv_try_rowmajor(bis_ctx *ctx, bis_mat *A);

bis_status bis_spmv_try_valdict(bis_ctx *ctx, bis_mat *A, bool consecutive_ok) {
    if (A->vd_state != 0) return BIS_OK;
    A->vd_state = -1;
    if (A->nnz == 0 || A->n_blocks == 0) return BIS_OK;
    int64_t ends[2];
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&ends[0], A->blk_nnz, 8, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&ends[1], A->blk_nnz + A->n_blocks, 8, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t s = ends[0], e = ends[1];
    if (e <= s) return BIS_OK;
    std::vector<unsigned long long> all;
    bool ok = false, diag = false;
    if (bis_status st = vd_collect(ctx, A, s, e, false, &all, &ok)) return st;
    if (ok && all.empty()) ok = false;
    const bool rm_possible = spmv_valdict_mode() >= 2 && A->max_row_nnz <= kRmMaxRow && A->n_cols < ((int64_t)1 << 29);
    if (!consecutive_ok && !rm_possible) return BIS_OK;
    if (!ok && rm_possible) {
        // few values apart from the diagonal?  (served by the lane-per-row form only: a lane knows its row)
        if (bis_status st = vd_collect(ctx, A, s, e, true, &all, &ok)) return st;
        diag = ok;
    }
    if (!ok) return BIS_OK;
    unsigned long long table[256] = {};
    std::copy(all.begin(), all.end(), table);
    A->vd_base = s & ~(int64_t)7; // whole 8-byte vectors of codes for the lane-per-row form
    const size_t n_code = (size_t)(e - A->vd_base) + 16;
    hipError_t he = hipMalloc(&A->vdict, sizeof table);
    if (he == hipSuccess) he = hipMalloc(&A->vcode, n_code);
    if (he == hipSuccess && diag) he = hipMalloc(&A->vdiag, sizeof(double) * (size_t)A->n_rows);
    if (he == hipSuccess) he = hipMemsetAsync(A->vcode, 0, n_code, ctx->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(A->vdict, table, sizeof table, hipMemcpyHostToDevice, ctx->stream);
    if (he == hipSuccess) {
        if (!diag) {
            const int grid = (int)std::min<int64_t>((e - A->vd_base + 1023) / 1024, (int64_t)ctx->n_cus * 32);
            hipLaunchKernelGGL(vd_encode_kernel, dim3(grid), dim3(256), 0, ctx->stream, A->val, s, e, A->vd_base, A->vdict, (int)all.size(), A->vcode);
        } else {
            const int grid = (int)std::min<int64_t>((A->n_rows + 255) / 256, (int64_t)ctx->n_cus * 32);
            if (A->rp64) hipLaunchKernelGGL(vd_encode_rows_kernel<int64_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int64_t *)A->row_ptr, A->col, A->val, A->n_rows, A->view_row0, A->vd_base, A->vdict, (int)all.size(), A->vcode, A->vdiag);
            else hipLaunchKernelGGL(vd_encode_rows_kernel<int32_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col, A->val, A->n_rows, A->view_row0, A->vd_base, A->vdict, (int)all.size(), A->vcode, A->vdiag);
        }
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream); // table[] lives on this stack frame
    if (he != hipSuccess) {
        bis_spmv_drop_valdict(A);
        A->vd_state = -1;
        BIS_HIP_CHECK(ctx, he);
    }
    A->vd_n = (int)all.size();
    A->vd_diag = diag;
    A->vd_rm_only = diag || !consecutive_ok;
    A->vd_state = 1;
    if (A->vd_rm_only) { // only the lane-per-row kernel can use it: does the matrix qualify?
        if (bis_status st = spmv_try_rowmajor(ctx, A)) return st;
        if (A->rm_state != 1) { bis_spmv_drop_valdict(A); A->vd_state = -1; }
    }
    return BIS_OK;
}

// default: select tree (see the tuning record in DESIGN.md section 4)
static int spmv_packed_mode() { return bis_opts().spmv_packed < 0 ? 1 : bis_opts().spmv_packed; }

bis_status bis_spmv_try_pack(bis_ctx *ctx, bis_mat *A, int t) {
    if (A->pk_state[t] != 0) return BIS_OK;
    A->pk_state[t] = -1;
    if (A->nnz == 0 || A->n_cols >= ((int64_t)1 << 29)) return BIS_OK;
    const int nb = t ? A->n_blocks_f : A->n_blocks;
    const int64_t *tab = t ? A->blkf_nnz : A->blk_nnz;
    int64_t ends[2];
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&ends[0], tab, 8, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(&ends[1], tab + nb, 8, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    A->pk_base[t] = ends[0] & ~(int64_t)3;
    const size_t n_pk = (size_t)(ends[1] - A->pk_base[t]) + 16;
    BIS_HIP_CHECK(ctx, hipMalloc(&A->pk[t], sizeof(uint16_t) * n_pk));
    BIS_HIP_CHECK(ctx, hipMalloc(&A->pk_seg[t], sizeof(int32_t) * (size_t)nb * kPk3Segs));
    int *status = (int *)ctx->counters + 44;
    // format 1: 8 windows of 8192 columns (select-tree decode); format 3: 32 windows of 2048 columns
    // (lane-permute decode) for matrices whose rows reach into many short column runs, e.g. after
    // a multi-colour reordering
    for (int kind = 1; kind <= 3; kind += 2) {
        if (kind == 3 && bis_opts().spmv_packed32 <= 0) break; // opt-in: measured no gain
        BIS_HIP_CHECK(ctx, hipMemsetAsync(status, 0, sizeof(int), ctx->stream));
        if (nb >= 256 && A->n_cols > 0) { // (the look before the build, see pk_probe_kernel)
            const int offbits = kind == 1 ? kPkOffBits : kPk3OffBits;
            const int n_words = (int)((((A->n_cols - 1) >> offbits) + 32) / 32);
            if ((size_t)n_words * 4 <= 48 * 1024) {
                if (kind == 1)
                    hipLaunchKernelGGL((pk_probe_kernel<kPkSegs, kPkOffBits>), dim3(64), dim3(256), (size_t)n_words * 4, ctx->stream, A->col, tab, nb, n_words, status);
                else
                    hipLaunchKernelGGL((pk_probe_kernel<kPk3Segs, kPk3OffBits>), dim3(64), dim3(256), (size_t)n_words * 4, ctx->stream, A->col, tab, nb, n_words, status);
                int hp = 0;
                BIS_HIP_CHECK(ctx, hipMemcpyAsync(&hp, status, sizeof hp, hipMemcpyDeviceToHost, ctx->stream));
                BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                if (hp) continue; // this window format cannot hold the matrix
            }
        }
        BIS_HIP_CHECK(ctx, hipMemsetAsync(A->pk[t], 0, sizeof(uint16_t) * n_pk, ctx->stream));
        if (kind == 1)
            hipLaunchKernelGGL((pk_build_kernel<kPkSegs, kPkOffBits>), dim3(nb), dim3(256), 0, ctx->stream, A->col, tab,
                               nb, A->pk_base[t], A->pk[t], A->pk_seg[t], status);
        else
            hipLaunchKernelGGL((pk_build_kernel<kPk3Segs, kPk3OffBits>), dim3(nb), dim3(256), 0, ctx->stream, A->col, tab,
                               nb, A->pk_base[t], A->pk[t], A->pk_seg[t], status);
        BIS_HIP_CHECK(ctx, hipGetLastError());
        int h = 0;
        BIS_HIP_CHECK(ctx, hipMemcpyAsync(&h, status, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (!h) { A->pk_state[t] = 1; A->pk_kind[t] = kind; break; }
    }
    if (A->pk_state[t] != 1) {
        hipFree(A->pk[t]); hipFree(A->pk_seg[t]);
        A->pk[t] = nullptr; A->pk_seg[t] = nullptr;
    }
    return BIS_OK;
}

// The packed stream of table t, built at the first use unless bis_mat_finalize
// already did (a cache on the matrix: the const handle of the SpMV entry points
// is cast away here only).  Identical tables (packed default) share stream 1.
static bis_status ensure_packed(bis_ctx *ctx, const bis_mat *A_c, int t, SpmvArgs *a) {
    bis_mat *A = const_cast<bis_mat *>(A_c);
    a->pk_mode = 0;
    a->wide = A->n_cols >= ((int64_t)1 << 29);
    if (a->wide || !spmv_packed_mode() || A->nnz == 0) return BIS_OK;
    if (A->chunk_nnz == A->chunk_f) t = 1;
    if (bis_status st = bis_spmv_try_pack(ctx, A, t)) return st;
    if (A->pk_state[t] == 1) {
        a->pk = A->pk[t]; a->pk_base = A->pk_base[t]; a->seg_base = A->pk_seg[t];
        a->col_max = (int)std::max<int64_t>(A->n_cols - 1, 0);
        a->pk_mode = A->pk_kind[t] == 3 ? 3 : (spmv_packed_mode() == 2 ? 2 : 1);
    }
    if (spmv_valdict_mode()) { // value dictionary: the consecutive kernel rides on the packed stream of the row-block table,
                               // the lane-per-row kernel brings its own
        if (bis_status st = bis_spmv_try_valdict(ctx, A, a->pk_mode == 1)) return st;
        if (A->vd_state == 1) { a->vcode = A->vcode; a->vd_base = A->vd_base; a->vdict = A->vdict; a->vd_rm_only = A->vd_rm_only; }
    }
    return BIS_OK;
}

// the block map's kernel argument and grid for nb logical blocks (shared with the forms of bis_spmv_sell.hip)
int bis_spmv_remap_arg(int nb) { return remap_arg_for((nb + 7) & ~7); }
int bis_spmv_grid(int nb) { return grid_for_map(nb, remap_arg_for((nb + 7) & ~7)); }

// x-window + sliced-ELL form of the dictionary kernel (bis_spmv_sell.hip) where the matrix qualifies; modes 0 and 1
static bool sellwin_wanted(const SpmvArgs &a) {
    return a.vcode && spmv_valdict_mode() >= 2 && bis_opts().spmv_sellwin != 0 && (bis_opts().spmv_variant < 0 || bis_opts().spmv_variant == 20);
}
static bis_status launch_sellwin(bis_ctx *ctx, const bis_mat *A, const SpmvArgs &a, const double *x, double *y, int mode,
                                 const double *w, double *partials, size_t partials_off, int *n_partials, bool *done) {
    *done = false;
    if (mode == 2 || !sellwin_wanted(a)) return BIS_OK;
    if (bis_status st = bis_spmv_sellwin_try(ctx, const_cast<bis_mat *>(A))) return st;
    const int nbr = bis_spmv_sellwin_blocks(A);
    if (!nbr) return BIS_OK;
    const int64_t n_slices = bis_spmv_sellwin_slices(A); // one partial of the fused dot per 64-row slice
    if (mode == 1 && partials_off + (size_t)n_slices > ctx->partials_cap) {
        ctx->err = "bis_spmv: partials buffer too small (internal)";
        return BIS_ERR_INVALID;
    }
    const int remap_arg = remap_arg_for((nbr + 7) & ~7);
    bis_prof_begin(ctx);
    bis_status st = bis_spmv_sellwin_launch(ctx, A, x, y, mode, w, mode == 1 ? partials + partials_off : partials, a.stop,
                                            remap_arg, grid_for_map(nbr, remap_arg));
    bis_prof_end(ctx);
    if (st != BIS_OK) return st;
    if (mode == 1 && n_partials) *n_partials = (int)n_slices;
    *done = true;
    return BIS_OK;
}

// window + sliced-ELL form with the 8-byte values streamed (bis_spmv_sell.hip, "win8"): for matrices the dictionary forms do not
// serve (arbitrary values, or spmv_valdict = 0), where the plan applies; modes 0 and 1
static bool win8_wanted() { return bis_opts().spmv_win8 != 0 && (bis_opts().spmv_variant < 0 || bis_opts().spmv_variant == 20); }
static bis_status launch_win8(bis_ctx *ctx, const bis_mat *A, const SpmvArgs &a, const double *x, double *y, int mode,
                              const double *w, double *partials, size_t partials_off, int *n_partials, bool *done) {
    *done = false;
    if (mode == 2 || !win8_wanted() || a.vcode) return BIS_OK; // (a matrix with a value dictionary keeps its dictionary kernels: 3 bytes per non-zero)
    if (bis_status st = bis_spmv_win8_try(ctx, const_cast<bis_mat *>(A))) return st;
    const int nbr = bis_spmv_win8_blocks(A);
    if (!nbr) return BIS_OK;
    const int64_t n_slices = bis_spmv_win8_partials(A); // (one partial of the fused dot per wave)
    if (mode == 1 && partials_off + (size_t)n_slices > ctx->partials_cap) {
        ctx->err = "bis_spmv: partials buffer too small (internal)";
        return BIS_ERR_INVALID;
    }
    const int remap_arg = remap_arg_for((nbr + 7) & ~7);
    bis_prof_begin(ctx);
    bis_status st = bis_spmv_win8_launch(ctx, A, x, y, mode, w, mode == 1 ? partials + partials_off : partials, a.stop,
                                         remap_arg, grid_for_map(nbr, remap_arg));
    bis_prof_end(ctx);
    if (st != BIS_OK) return st;
    if (mode == 1 && n_partials) *n_partials = (int)n_slices;
    *done = true;
    return BIS_OK;
}

// lane-per-row form of the dictionary kernel where the matrix qualifies (*done tells); mode 0 / 1 / 2 as in SpmvArgs
static bis_status launch_rowmajor(bis_ctx *ctx, const bis_mat *A, const SpmvArgs &a, const double *x, double *y, int mode,
                                  const double *w, double *partials, size_t partials_off, int *n_partials, bool *done) {
    *done = false;
    if (!(a.vcode && spmv_valdict_mode() >= 2 && (bis_opts().spmv_variant < 0 || bis_opts().spmv_variant == 20))) return BIS_OK;
    if (bis_status st = spmv_try_rowmajor(ctx, const_cast<bis_mat *>(A))) return st;
    if (A->rm_state != 1) return BIS_OK;
    const int nbr = A->rm_blocks, nbr8 = (nbr + 7) & ~7;
    if (mode == 1 && partials_off + (size_t)nbr * 4 > ctx->partials_cap) {
        ctx->err = "bis_spmv: partials buffer too small (internal)";
        return BIS_ERR_INVALID;
    }
    const int remap_arg = remap_arg_for(nbr8);
    const int grid = grid_for_map(nbr, remap_arg);
    const int code_cap = kRmRows * std::max(A->max_row_nnz, 1) + 16; // positions: a block's codes + the 8-alignment slack on both sides
    const size_t lds = 3 * (size_t)code_cap;
    double *pp = mode == 1 ? partials + partials_off : partials;
    if (mode != 2) bis_prof_begin(ctx);
#define BIS_RM_LAUNCH3(RP, MODE, DIAG, W32)                                                                            \
    hipLaunchKernelGGL((spmv_rowmajor_vd_kernel<RP, MODE, DIAG, W32>), dim3(grid), dim3(256), lds, ctx->stream, (const RP *)A->row_ptr, x, y, \
                       A->n_rows, nbr, remap_arg, w, pp, A->rm_pk, A->rm_base, A->rm_seg, a.stop, A->vcode, A->vd_base, A->vdict, code_cap, \
                       A->vdiag)
#define BIS_RM_LAUNCH2(RP, MODE, DIAG) do { if (A->rm_kind == 3) BIS_RM_LAUNCH3(RP, MODE, DIAG, true); else BIS_RM_LAUNCH3(RP, MODE, DIAG, false); } while (0)
#define BIS_RM_LAUNCH(RP, MODE) do { if (A->vd_diag) BIS_RM_LAUNCH2(RP, MODE, true); else BIS_RM_LAUNCH2(RP, MODE, false); } while (0)
    if (A->rp64) { if (mode == 2) BIS_RM_LAUNCH(int64_t, 2); else if (mode == 1) BIS_RM_LAUNCH(int64_t, 1); else BIS_RM_LAUNCH(int64_t, 0); }
    else { if (mode == 2) BIS_RM_LAUNCH(int32_t, 2); else if (mode == 1) BIS_RM_LAUNCH(int32_t, 1); else BIS_RM_LAUNCH(int32_t, 0); }
#undef BIS_RM_LAUNCH
#undef BIS_RM_LAUNCH2
#undef BIS_RM_LAUNCH3
    if (mode != 2) bis_prof_end(ctx);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    if (mode == 1 && n_partials) *n_partials = nbr * 4;
    *done = true;
    return BIS_OK;
}

// ---- column slabs (bis_spmv_slab.hip): K passes of the row-block kernel, each gathering from an x slice that fits the L2 ----

void bis_spmv_colslab_drop(bis_mat *A) {
    if (A->colslabs) {
        for (bis_mat *B : *A->colslabs) { // (no context at hand: what bis_mat_destroy does for a matrix that owns its arrays)
            bis_mat_free_meta(B);
            hipFree(B->row_ptr); hipFree(B->col); hipFree(B->val);
            delete B;
        }
        delete A->colslabs;
        A->colslabs = nullptr;
    }
    A->cs_state = 0;
    A->cs_trial_ms[0] = A->cs_trial_ms[1] = 0.0;
}

// one pass of the row-block kernel over matrix B (A itself, or one of its slabs)
static bis_status rowblock_pass(bis_ctx *ctx, const bis_mat *B, const double *x, double *y, const double *w, size_t partials_off,
                                int acc_y, const int *stop, int *n_partials) {
    const bool use_f = w != nullptr;
    const int64_t lds_doubles = (int64_t)(use_f ? B->chunk_f : B->chunk_nnz) + B->max_row_nnz + 8;
    const int nb = use_f ? B->n_blocks_f : B->n_blocks, nb8 = (nb + 7) & ~7;
    if (w && partials_off + (size_t)nb * 4 > ctx->partials_cap) { ctx->err = "bis_spmv: partials buffer too small (internal)"; return BIS_ERR_INVALID; }
    SpmvArgs a{B->row_ptr, B->col, B->val, x, y, use_f ? B->blkf_row : B->blk_row, use_f ? B->blkf_nnz : B->blk_nnz, nb, nb8, w,
               ctx->partials + partials_off, sizeof(double) * (size_t)lds_doubles, ctx->stream, w ? 1 : 0};
    a.n_cus = ctx->n_cus;
    a.remap = bis_opts().spmv_xcd_remap == 1;
    a.remap_arg = remap_arg_for(nb8);
    a.grid = grid_for_map(nb, a.remap_arg);
    if (bis_status st = ensure_packed(ctx, B, use_f ? 1 : 0, &a)) return st;
    a.vcode = nullptr; // (the plain row-block kernel: the value-dictionary kernel has no accumulating form)
    a.stop = stop;
    a.acc_y = acc_y;
    const bool ok = launch_by_id<int32_t>(spmv_variant(a), a);
    if (!ok) { ctx->err = "bis_spmv: unknown BIS_SPMV_VARIANT"; return BIS_ERR_INVALID; }
    if (w && n_partials) *n_partials = nb * (fused_threads(spmv_variant(a)) / 64);
    return BIS_OK;
}

static bis_status colslab_passes(bis_ctx *ctx, const bis_mat *A, const double *x, double *y, const double *w, size_t partials_off,
                                 const int *stop, int *n_partials) {
    const std::vector<bis_mat *> &S = *A->colslabs;
    for (size_t k = 0; k < S.size(); ++k) {
        const bool last = k + 1 == S.size();
        if (bis_status st = rowblock_pass(ctx, S[k], x, y, last ? w : nullptr, partials_off, k > 0 ? 1 : 0, stop, last ? n_partials : nullptr)) return st;
    }
    return BIS_OK;
}

// Builds the slabs where the plan applies: 32-bit row pointers, not a row view, rows of at most the LDS budget, a column stream
// that did NOT pack (a matrix whose row blocks each touch at most 8 windows of 8192 columns has the locality the slabs would
// make), an x of more than 6 MiB, ascending rows.  Default: K = x's bytes / 2 MiB (2..32) and a trial -- three timed launches of
// the one pass and of the K passes on a scratch x -- keeps the slabs only where they are at least 15 % faster.
static bis_status colslab_try(bis_ctx *ctx, bis_mat *A, bool packed) {
    if (A->cs_state != 0) return BIS_OK;
    A->cs_state = -1;
    const int opt = bis_opts().spmv_colslab;
    const bool forced = opt >= 2;
    if (opt == 0 || A->rp64 || A->view || A->n_rows == 0 || A->nnz == 0) return BIS_OK;
    if (!forced && (packed || 8 * A->n_cols <= (int64_t)6 << 20 || A->nnz < ((int64_t)1 << 22))) return BIS_OK;
    if (sizeof(double) * (size_t)((int64_t)A->chunk_f + A->max_row_nnz + 8) > 64 * 1024) return BIS_OK;
    int K = forced ? std::min(opt, 32) : (int)std::min<int64_t>(32, std::max<int64_t>(2, (8 * A->n_cols + ((int64_t)2 << 20) - 1) / ((int64_t)2 << 20)));
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (int64_t)free_b < 16 * A->nnz + ((int64_t)8 << 30)) { (void)hipGetLastError(); return BIS_OK; }
    auto *slabs = new std::vector<bis_mat *>();
    bool ok = false;
    bis_status st = bis_spmv_colslab_build(ctx, A, K, *slabs, &ok);
    if (st != BIS_OK || !ok) { delete slabs; return st; }
    A->colslabs = slabs;
    if (!forced) { // the trial
        double *xs = nullptr, *ys = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool good = hipMalloc(&xs, sizeof(double) * (size_t)A->n_cols) == hipSuccess && hipMalloc(&ys, sizeof(double) * (size_t)A->n_rows) == hipSuccess &&
                    hipMemsetAsync(xs, 0, sizeof(double) * (size_t)A->n_cols, ctx->stream) == hipSuccess &&
                    hipEventCreate(&ev[0]) == hipSuccess && hipEventCreate(&ev[1]) == hipSuccess;
        float ms[2] = {0.f, 0.f};
        for (int which = 0; which < 2 && good && st == BIS_OK; ++which) {
            for (int rep = 0; rep < 5 && st == BIS_OK; ++rep) {
                if (rep == 2) good = good && hipEventRecord(ev[0], ctx->stream) == hipSuccess;
                st = which ? colslab_passes(ctx, A, xs, ys, nullptr, 0, nullptr, nullptr) : rowblock_pass(ctx, A, xs, ys, nullptr, 0, 0, nullptr, nullptr);
            }
            good = good && hipEventRecord(ev[1], ctx->stream) == hipSuccess && hipEventSynchronize(ev[1]) == hipSuccess &&
                   hipEventElapsedTime(&ms[which], ev[0], ev[1]) == hipSuccess;
        }
        if (ev[0]) hipEventDestroy(ev[0]);
        if (ev[1]) hipEventDestroy(ev[1]);
        hipFree(xs); hipFree(ys);
        (void)hipGetLastError();
        A->cs_trial_ms[0] = ms[0] / 3.0;
        A->cs_trial_ms[1] = ms[1] / 3.0;
        if (st != BIS_OK || !good || !(ms[1] < 0.85f * ms[0])) {
            const double keep[2] = {A->cs_trial_ms[0], A->cs_trial_ms[1]};
            bis_spmv_colslab_drop(A);
            A->cs_state = -1;
            A->cs_trial_ms[0] = keep[0]; A->cs_trial_ms[1] = keep[1];
            return st;
        }
    }
    A->cs_state = 1;
    return BIS_OK;
}

static bool colslab_wanted(const SpmvArgs &a) {
    return bis_opts().spmv_colslab != 0 && !a.vcode && !a.wide && (bis_opts().spmv_variant < 0 || bis_opts().spmv_variant == 20 || bis_opts().spmv_variant == 41);
}
static bis_status launch_colslab(bis_ctx *ctx, const bis_mat *A, const SpmvArgs &a, const double *x, double *y, const double *w,
                                 size_t partials_off, int *n_partials, bool *done) {
    *done = false;
    if (!colslab_wanted(a)) return BIS_OK;
    if (bis_status st = colslab_try(ctx, const_cast<bis_mat *>(A), a.pk_mode != 0)) return st;
    if (A->cs_state != 1) return BIS_OK;
    bis_prof_begin(ctx);
    bis_status st = colslab_passes(ctx, A, x, y, w, partials_off, a.stop, n_partials);
    bis_prof_end(ctx);
    if (st != BIS_OK) return st;
    BIS_HIP_CHECK(ctx, hipGetLastError());
    *done = true;
    return BIS_OK;
}

// internal: y = A x, optionally partials[b] = sum_{r in block b} y[r]*w[r]
// (n_partials returns the number of partials written; 0 if not fused).
bis_status bis_spmv_launch(bis_ctx *ctx, const bis_mat *A, const double *x, double *y,
                           const double *w, int *n_partials, size_t partials_off) {
    if (n_partials) *n_partials = 0;
    if (A->n_rows == 0) return BIS_OK;
    const bool use_f = w != nullptr && !(A->win_ok && spmv_window_mode()); // fused epilogue: its own table
    const int64_t lds_doubles = (int64_t)(use_f ? A->chunk_f : A->chunk_nnz) + A->max_row_nnz + 8;
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
    const int nb = use_f ? A->n_blocks_f : A->n_blocks, nb8 = (nb + 7) & ~7;
    if (w && partials_off + (size_t)nb * 4 > ctx->partials_cap) {
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
                       bis_opts().spmv_xcd_remap > 0 ? nb8 : -1, w, ctx->partials + partials_off)
        if (A->rp64) { if (w) BIS_WIN_LAUNCH(int64_t, true); else BIS_WIN_LAUNCH(int64_t, false); }
        else { if (w) BIS_WIN_LAUNCH(int32_t, true); else BIS_WIN_LAUNCH(int32_t, false); }
#undef BIS_WIN_LAUNCH
        bis_prof_end(ctx);
        BIS_HIP_CHECK(ctx, hipGetLastError());
        if (w && n_partials) *n_partials = nb * 4; // 256-thread workgroups: 4 waves
        return BIS_OK;
    }
    SpmvArgs a{A->row_ptr, A->col, A->val, x, y, use_f ? A->blkf_row : A->blk_row,
               use_f ? A->blkf_nnz : A->blk_nnz, nb, nb8, w,
               ctx->partials + partials_off, lds_bytes,
               ctx->stream, w ? 1 : 0};
    a.n_cus = ctx->n_cus;
    a.remap = bis_opts().spmv_xcd_remap == 1;
    a.remap_arg = remap_arg_for(nb8);
    a.grid = grid_for_map(nb, a.remap_arg);
    if (bis_status st = ensure_packed(ctx, A, use_f ? 1 : 0, &a)) return st;
    a.stop = ctx->spmv_stop; // a device schedule (bis_cg_iterate, bis_stat_iterate) is enqueuing: no-op once its stop flag is set
    {
        bool done = false;
        if (bis_status st = launch_sellwin(ctx, A, a, x, y, w ? 1 : 0, w, ctx->partials, partials_off, n_partials, &done)) return st;
        if (done) return BIS_OK;
    }
    {
        bool done = false;
        if (bis_status st = launch_rowmajor(ctx, A, a, x, y, w ? 1 : 0, w, ctx->partials, partials_off, n_partials, &done)) return st;
        if (done) return BIS_OK;
    }
    {
        bool done = false;
        if (bis_status st = launch_win8(ctx, A, a, x, y, w ? 1 : 0, w, ctx->partials, partials_off, n_partials, &done)) return st;
        if (done) return BIS_OK;
    }
    {
        bool done = false;
        if (bis_status st = launch_colslab(ctx, A, a, x, y, w, partials_off, n_partials, &done)) return st;
        if (done) return BIS_OK;
    }
    if (bis_opts().spmv_lds_pad > 0) a.lds_bytes += (size_t)bis_opts().spmv_lds_pad;
    bis_prof_begin(ctx);
    const bool ok = A->rp64 ? launch_by_id<int64_t>(spmv_variant(a), a)
                            : launch_by_id<int32_t>(spmv_variant(a), a);
    bis_prof_end(ctx);
    if (!ok) { ctx->err = "bis_spmv: unknown BIS_SPMV_VARIANT"; return BIS_ERR_INVALID; }
    BIS_HIP_CHECK(ctx, hipGetLastError());
    if (w && n_partials) *n_partials = nb * (fused_threads(spmv_variant(a)) / 64);
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
    a.n_cus = ctx->n_cus;
    a.remap = bis_opts().spmv_xcd_remap == 1;
    a.remap_arg = remap_arg_for(nb8);
    a.grid = grid_for_map(nb, a.remap_arg);
    if (bis_status st = ensure_packed(ctx, T, 0, &a)) return st;
    a.stop = ctx->spmv_stop;
    {
        bool done = false;
        if (bis_status st = launch_rowmajor(ctx, T, a, x, y, 2, b, const_cast<double *>(D), 0, nullptr, &done)) return st;
        if (done) return BIS_OK;
    }
    const bool ok = T->rp64 ? launch_by_id<int64_t>(spmv_variant(a), a) : launch_by_id<int32_t>(spmv_variant(a), a);
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

bis_status bis_mat_spmv_stream_info(bis_ctx *ctx, const bis_mat *A, int *col_bytes, int *val_bytes, int *n_dict, int *form) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A, "bis_mat_spmv_stream_info: bad arguments");
    SpmvArgs a{};
    const int64_t lds_doubles = (int64_t)A->chunk_nnz + A->max_row_nnz + 8;
    if (A->n_rows > 0 && sizeof(double) * (size_t)lds_doubles <= 64 * 1024 && !(A->win_ok && spmv_window_mode()))
        if (bis_status st = ensure_packed(ctx, A, 0, &a)) return st;
    int f = 0;
    if (sellwin_wanted(a)) {
        if (bis_status st = bis_spmv_sellwin_try(ctx, const_cast<bis_mat *>(A))) return st;
        if (bis_spmv_sellwin_blocks(A)) f = A->vd_diag ? 5 : 4;
    }
    if (!f && a.vcode && spmv_valdict_mode() >= 2 && (bis_opts().spmv_variant < 0 || bis_opts().spmv_variant == 20)) {
        if (bis_status st = spmv_try_rowmajor(ctx, const_cast<bis_mat *>(A))) return st;
        if (A->rm_state == 1) f = A->vd_diag ? 3 : 2;
    }
    if (!f && a.vcode && !a.vd_rm_only && spmv_variant(a) == 20 && a.pk_mode == 1) f = 1;
    if (!f && !a.vcode && win8_wanted() && A->n_rows > 0) { // form 6: window + sliced ELL, 8-byte values + 2-byte window slots
        if (bis_status st = bis_spmv_win8_try(ctx, const_cast<bis_mat *>(A))) return st;
        if (bis_spmv_win8_blocks(A)) f = 6;
    }
    if (!f && colslab_wanted(a) && A->n_rows > 0 && sizeof(double) * (size_t)lds_doubles <= 64 * 1024) { // form 7: K column slabs, each a pass of the row-block kernel
        if (bis_status st = colslab_try(ctx, const_cast<bis_mat *>(A), a.pk_mode != 0)) return st;
        if (A->cs_state == 1) {
            bool all_packed = true;
            for (const bis_mat *B : *A->colslabs) {
                SpmvArgs b{};
                if (bis_status st = ensure_packed(ctx, B, 0, &b)) return st;
                all_packed = all_packed && (b.pk_mode != 0 || B->nnz == 0);
            }
            if (col_bytes) *col_bytes = all_packed ? 2 : 4;
            if (val_bytes) *val_bytes = 8;
            if (n_dict) *n_dict = (int)A->colslabs->size(); // (the number of slabs)
            if (form) *form = 7;
            return BIS_OK;
        }
    }
    if (f == 6) {
        if (col_bytes) *col_bytes = 2;
        if (val_bytes) *val_bytes = 8;
        if (n_dict) *n_dict = 0;
        if (form) *form = 6;
        return BIS_OK;
    }
    if (col_bytes) *col_bytes = (f >= 4 && bis_spmv_sellwin_format(A) == 4) ? 0 : (f >= 4 && bis_spmv_sellwin_format(A) == 3) ? 1 : ((f >= 2 || a.pk_mode) ? 2 : 4); // 1: one byte per non-zero, the index of its (column - row, value) pair; 0: a 32-bit mask of pairs per ROW
    if (val_bytes) *val_bytes = f ? (f >= 4 && bis_spmv_sellwin_format(A) >= 2 ? 0 : 1) : 8; // 0: the value index shares the column code
    if (n_dict) *n_dict = f ? A->vd_n : 0;
    if (form) *form = f;
    return BIS_OK;
}

bis_status bis_mat_spmv_streamed_bytes(bis_ctx *ctx, const bis_mat *A, int64_t *bytes) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, A && bytes, "bis_mat_spmv_streamed_bytes: bad arguments");
    int col_b = 4, val_b = 8, n_dict = 0, form = 0;
    if (bis_status st = bis_mat_spmv_stream_info(ctx, A, &col_b, &val_b, &n_dict, &form)) return st;
    const int64_t rp = A->rp64 ? 8 : 4;
    int64_t b = 8 * A->n_cols + 8 * A->n_rows; // x once, y once
    if (form == 7) { // the slabs' CRS copies, and y read and written again by every pass after the first
        for (const bis_mat *B : *A->colslabs)
            b += (int64_t)(col_b + val_b) * B->nnz + rp * (B->n_rows + 1) + (int64_t)B->n_blocks * (12 + (col_b == 2 ? 32 : 0));
        b += 16 * A->n_rows * ((int64_t)A->colslabs->size() - 1);
    } else if (form == 6) {
        b += bis_spmv_win8_bytes(A);
    } else if (form >= 4) {
        b += bis_spmv_sellwin_bytes(A);
    } else if (form >= 2) {
        b += 3 * A->nnz + rp * (A->n_rows + 1) + (int64_t)A->rm_blocks * (A->rm_kind == 3 ? 128 : 32) + 2048;
    } else {
        b += (int64_t)(col_b + val_b) * A->nnz + rp * (A->n_rows + 1) + (int64_t)A->n_blocks * (12 + (col_b == 2 ? 32 : 0)) + (form == 1 ? 2048 : 0);
    }
    if (form == 3 || form == 5) b += 8 * A->n_rows; // the per-row diagonal values
    *bytes = b;
    return BIS_OK;
}

void bis_mat_colslab_info(const bis_mat *A, int *slabs, double *one_pass_ms, double *slab_passes_ms) {
    if (slabs) *slabs = (A && A->cs_state == 1 && A->colslabs) ? (int)A->colslabs->size() : 0;
    if (one_pass_ms) *one_pass_ms = A ? A->cs_trial_ms[0] : 0.0;
    if (slab_passes_ms) *slab_passes_ms = A ? A->cs_trial_ms[1] : 0.0;
}

bis_status bis_compute_residual(bis_ctx *ctx, const bis_mat *A, const double *x, const double *b,
                                double *res, double *tmp) {
    // kernels.hpp:155-162: tmp = A x ; res = b - tmp
    bis_status st = bis_spmv(ctx, A, x, tmp);
    if (st != BIS_OK) return st;
    return bis_subtract_vectors(ctx, res, b, tmp, A->n_rows, 1.0);
}

} // extern "C"
