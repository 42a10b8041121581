// bis_spmv_sell.hip -- CRS SpMV (kernels.hpp:22-42 of the reference), dictionary form with the block's x window in
// LDS and the codes in a sliced-ELL stream.  gfx950 only.
//
// The lane-per-row dictionary kernel (bis_spmv.hip, spmv_rowmajor_vd_kernel) is bound by the texture addresser: one
// 8-byte gather per non-zero, plus the LDS round trip that turns the CRS-ordered code stream into per-lane order.
// This form removes both:
//   * window: the x entries a block of 256 consecutive rows reads are, for a banded / stencil matrix, a few
//     contiguous runs (HPCG: 9 grid lines).  The plan records the runs in units of 8-column granules (64 B); the
//     kernel copies them into LDS with coalesced 16-byte loads, every x entry once per block, and a column code is
//     the LDS byte offset of its entry -- no window decode, no gather through the texture path.
//   * sliced ELL: the codes of a slice (64 consecutive rows = one wave) are stored chunk-major -- chunk c holds
//     entries 4c..4c+3 of every row of the slice, 12 bytes per lane (4 x 16-bit column code, 4 x 8-bit value code)
//     -- so a lane loads its own codes with one coalesced 12-byte load per 4 non-zeros: no staging, no row_ptr.
//     Where the table (dictionary + padding value [+ the per-row-diagonal marker]) has at most 8 entries -- every
//     constant-coefficient stencil -- a non-zero is ONE 16-bit code, window slot : 13 | table index : 3, 8 bytes per
//     lane and chunk: 2 streamed bytes per non-zero.
//     Where the matrix has at most 253 distinct (column - row, value) pairs -- a stencil on a structured grid has as
//     many as it has points -- and, inside every block, the columns of each pair run through the window in step with
//     the rows (window slot = base of the pair in this block + row: true when they lie in one run), a non-zero is ONE
//     BYTE: the index of its pair; the block brings its bases (2 bytes per pair).  1 streamed byte per non-zero.
//   * a block is 256 R rows (R = 1, 2, 4 rows per lane, one 64-row slice after the other): the larger R, the fewer
//     times an x entry is copied into some block's window (HPCG: 9 x at R = 1, 6 x at R = 2), at a larger window.
//     A slice has as many chunks as its longest row needs; shorter rows are padded with an entry that is neutral in
//     IEEE arithmetic whatever the accumulator holds: value 1.0 times an LDS slot holding -0.0 (acc + -0.0 == acc
//     for every acc, including -0.0, infinities and NaN).
// Entry j of a row is the j-th entry of the CRS row, value and product are the CRS ones, the sum runs in CRS order
// with a rounding after the product and after the sum as in the other kernels: y is bit-identical to theirs.
// The CRS arrays stay authoritative; the form is built lazily on the device and dropped when values change.
// A matrix some block of which needs more than 32 runs or more than 60 KiB of window, or whose padding exceeds 30 %,
// keeps the gather kernels.
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "bis_internal.hpp"

struct bis_sellwin {
    int n_blocks = 0;
    int64_t n_slices = 0, total_chunks = 0;
    int32_t *hdr = nullptr;          // [n_blocks * 64]: words 0..31 first granule of run k, words 32..63 (rank of the run's first granule in the window) | (granules << 16)
    int64_t *slice_chunk0 = nullptr; // [n_slices + 1]
    int32_t *own_rank = nullptr;     // [n_blocks] rank (in the window) of the granule of the block's first row's OWN column (row + view_row0) when the
                                     // own columns of all the block's rows lie in the window side by side, else -1: the fused dot then takes w from LDS
    uint32_t *codes = nullptr;       // [(total_chunks + 1) * 64 * words], words = 3, 3, 2, 1 for fmt 0..3
    double *dict = nullptr;          // [256] the matrix' dictionary plus the padding value 1.0
    int max_gran = 0;                // largest window of a block, in granules
    int fmt = 0;                     // 0: 12-byte chunks, value byte = table index; 1: value byte = 8 * index (<= 32 entries);
                                     // 2: 8-byte chunks, 16-bit codes slot : 13 | index : 3 (<= 8 entries);
                                     // 3: 4-byte chunks, one byte per non-zero = index of its (column - row, value) pair
                                     // 4: no chunks: codes[row] = 32-bit mask of the row's pairs (see spmv_sellmask_kernel)
    int uniform_chunks = 0;          // > 0: every slice has this many chunks (short slices padded up when that costs < 3 % of the stream)
    int n_pairs = 0, pair_stride = 0, diag_pair = -1; // fmt 3: pairs of the matrix, int16 words per block in blk_base, the per-row-diagonal pair
    int16_t *blk_base = nullptr;     // fmt 3: [n_blocks * pair_stride] window slot of pair e's column for the block's first row
    unsigned long long *pair_key = nullptr; // fmt 3 / 4: [256] the pairs, ascending (sw_pair_key: column - row, then value code)
    uint16_t *row_of = nullptr;      // win8: [n_blocks * 256 R] row (in its block) of every position of the block's length order
    void *own_codes = nullptr;       // win8: the library's own stream while a debugging caller has redirected `codes` (bis_mat_win8_debug_stream)
    int tune_trials = 0;             // win8: placement tuning at build time (re-allocations tried), the kernel's time on the first
    double tune_first_ms = 0.0, tune_kept_ms = 0.0; // allocation and on the one kept
    int R = 1;                       // rows per lane: a block is 256 R rows
    bool diag = false;               // one value code stands for the row's own diagonal value (vdiag)
    int pad_idx = 0, diag_idx = 0;
};

namespace {

constexpr int kSwRows = 256;
constexpr int kSwRuns = 32;
constexpr int kSwRunGran = 64; // granules per run at most
constexpr int kSwHash = 4096;
constexpr int kSwMaxGran = 940; // window <= (2 + 8 * 940) * 8 = 60176 bytes (list[1024] in the plan kernel, 13-bit slots in the joint codes)
constexpr int kSwDefaultR = 2;
constexpr int kSwMaxPairs = 253; // fmt 3: pairs of a matrix at most (the padding takes the next index)
// ... and with the tables in front of it the workgroup stays within 64 KiB of LDS
inline int sw_gran_cap(int R) { return R == 4 ? kSwMaxGran : std::min(kSwMaxGran, (65536 - (4096 + 4096 * R) - 16) / 64); } // (the largest table area: fmt 3 with per-row diagonals; R = 4 is the row-mask form only: no tables)

// a (column - row, value code) pair as one key: ascending keys = ascending (signed) offsets, so a row with ascending columns runs
// through the sorted list front to back
__device__ __forceinline__ unsigned long long sw_pair_key(int32_t offset, unsigned vcode) {
    return ((unsigned long long)((unsigned)offset ^ 0x80000000u) << 8) | vcode;
}

__device__ __forceinline__ int sw_wave_max(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

// One workgroup per block of 256 rows: the 8-column granules of x the block reads (LDS hash set, bitonic sort), cut
// into runs of consecutive granules; chunks per slice.  status[0] = 1: not representable; status[1] = max granules.
template <typename RP>
__global__ __launch_bounds__(256) void sw_plan_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                      int64_t n_rows, int R, int max_gran, int64_t row0, int32_t *__restrict__ hdr,
                                                      int32_t *__restrict__ slice_chunks, int32_t *__restrict__ own_rank, int *status) {
    __shared__ int table[kSwHash];
    __shared__ int list[1024];
    __shared__ int cnt, n_runs, failed;
    __shared__ int run_g0[kSwRuns], run_rank[kSwRuns], srt_g0[kSwRuns], srt_rank[kSwRuns + 1];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t r0 = (int64_t)b * kSwRows * R;
    const int rows = (int)min((int64_t)kSwRows * R, n_rows - r0);
    for (int i = tid; i < kSwHash; i += 256) table[i] = -1;
    if (tid == 0) { cnt = 0; n_runs = 0; failed = 0; }
    __syncthreads();
    for (int r = 0; r < R; ++r) { // wave w owns the R consecutive slices w R .. w R + R - 1 of the block
        const int sl = (tid >> 6) * R + r, lr = sl * 64 + (tid & 63);
        int len = 0;
        if (lr < rows) len = (int)((int64_t)row_ptr[r0 + lr + 1] - (int64_t)row_ptr[r0 + lr]);
        const int m = sw_wave_max(len);
        if ((tid & 63) == 0) { slice_chunks[(size_t)b * 4 * R + sl] = (m + 3) >> 2; atomicMax(&status[5], (m + 3) >> 2); }
    }
    const int64_t s = (int64_t)row_ptr[r0], e = (int64_t)row_ptr[r0 + rows];
    for (int64_t k = s + tid; k < e; k += 256) {
        if (__hip_atomic_load(&failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        const int g = col[k] >> 3;
        unsigned h = ((unsigned)g * 2654435761u) >> 20;
        for (;;) {
            const int cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (cur == g) break;
            if (cur == -1) {
                const int old = atomicCAS(&table[h], -1, g);
                if (old == -1) {
                    if (atomicAdd(&cnt, 1) >= max_gran) atomicExch(&failed, 1);
                    break;
                }
                if (old == g) break;
            }
            h = (h + 1) & (kSwHash - 1);
        }
    }
    __syncthreads();
    const int n = cnt;
    if (failed || n > max_gran) {
        if (tid == 0) atomicExch(&status[0], 1);
        return;
    }
    __syncthreads();
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (int i = tid; i < kSwHash; i += 256) {
        const int g = table[i];
        if (g != -1) list[atomicAdd(&cnt, 1)] = g;
    }
    int P = 2;
    while (P < n) P <<= 1;
    __syncthreads();
    for (int i = n + tid; i < P; i += 256) list[i] = INT32_MAX;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += 256) {
                const int o = i ^ j;
                if (o > i) {
                    const int a = list[i], c = list[o];
                    if ((a > c) == ((i & k) == 0)) { list[i] = c; list[o] = a; }
                }
            }
            __syncthreads();
        }
    if (tid == 0) { // are the granules of the block's own columns (row + row0) all in the window, side by side?
        const int64_t c_first = r0 + row0, c_last = c_first + rows - 1;
        int own = -1;
        if ((c_first & 7) == 0) {
            const int g_first = (int)(c_first >> 3), g_last = (int)(c_last >> 3);
            int lo = 0, hi = n - 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (list[mid] < g_first) lo = mid + 1; else hi = mid; }
            if (n > 0 && list[lo] == g_first && lo + (g_last - g_first) < n && list[lo + (g_last - g_first)] == g_last) own = lo;
        }
        own_rank[b] = own;
    }
    for (int i = tid; i < n; i += 256)
        if (i == 0 || list[i] != list[i - 1] + 1) {
            const int idx = atomicAdd(&n_runs, 1);
            if (idx < kSwRuns) { run_g0[idx] = list[i]; run_rank[idx] = i; }
        }
    __syncthreads();
    const int nr = n_runs;
    if (nr > kSwRuns) {
        if (tid == 0) atomicExch(&status[0], 1);
        return;
    }
    if (tid < nr) {
        const int my = run_rank[tid];
        int pos = 0;
        for (int m = 0; m < nr; ++m) pos += run_rank[m] < my;
        srt_g0[pos] = run_g0[tid];
        srt_rank[pos] = my;
    }
    if (tid == 0) srt_rank[nr] = n;
    __syncthreads();
    // runs of at most 64 granules (256 pieces of 16 bytes: one piece per thread of the SpMV's workgroup)
    if (tid == 0) {
        int m = 0;
        for (int k = 0; k < nr; ++k) {
            const int len = srt_rank[k + 1] - srt_rank[k];
            for (int o = 0; o < len; o += kSwRunGran, ++m)
                if (m < kSwRuns) {
                    run_g0[m] = srt_g0[k] + o;
                    run_rank[m] = (srt_rank[k] + o) | (min(kSwRunGran, len - o) << 16);
                }
        }
        n_runs = m;
    }
    __syncthreads();
    const int nr2 = n_runs;
    if (nr2 > kSwRuns) {
        if (tid == 0) atomicExch(&status[0], 1);
        return;
    }
    if (tid < 64) {
        const int k = tid & 31;
        int word = 0;
        if (k < nr2) word = tid < 32 ? run_g0[k] : run_rank[k];
        hdr[(size_t)b * 64 + tid] = word;
    }
    if (tid == 0) atomicMax(&status[1], n);
}

__global__ __launch_bounds__(256) void sw_uniform_kernel(int64_t *__restrict__ chunk0, int64_t n, int64_t per) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) chunk0[i] = i * per;
}

__global__ __launch_bounds__(256) void sw_widen_kernel(const int32_t *__restrict__ in, int64_t *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// One workgroup per block: a lane writes the chunks of its R rows.
template <typename RP>
__global__ __launch_bounds__(256) void sw_fill_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                      const uint8_t *__restrict__ vcode, int64_t vd_base, int64_t n_rows, int R, int fmt,
                                                      const int32_t *__restrict__ hdr, const int64_t *__restrict__ slice_chunk0,
                                                      uint32_t *__restrict__ codes, int pad_idx, int diag_idx) {
    __shared__ int g0s[kSwRuns], rk[kSwRuns];
    __shared__ int nr_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < kSwRuns) {
        g0s[tid] = hdr[(size_t)b * 64 + tid];
        const int w2 = hdr[(size_t)b * 64 + 32 + tid];
        rk[tid] = w2 & 0xffff;
        const unsigned long long m = __ballot((w2 >> 16) != 0);
        if (tid == 0) nr_s = __popcll(m);
    }
    __syncthreads();
    const int nr = nr_s;
    const int words = fmt == 2 ? 2 : 3;
    for (int rr = 0; rr < R; ++rr) {
        const int64_t slice = ((int64_t)b * 4 + wv) * R + rr;
        const int64_t r = slice * 64 + lane;
        int64_t rs = 0;
        int len = 0;
        if (r < n_rows) {
            rs = (int64_t)row_ptr[r];
            len = (int)((int64_t)row_ptr[r + 1] - rs);
        }
        const int64_t c0 = slice_chunk0[slice];
        const int nch = (int)(slice_chunk0[slice + 1] - c0);
        for (int c = 0; c < nch; ++c) {
            unsigned cc[4], vv[4]; // window slot, table index
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 4 * c + q;
                cc[q] = 0;
                vv[q] = (unsigned)pad_idx;
                if (j < len) {
                    const int ci = col[rs + j];
                    const int g = ci >> 3;
                    int lo = 0, hi = nr - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (g0s[mid] <= g) lo = mid; else hi = mid - 1;
                    }
                    cc[q] = (unsigned)(2 + (rk[lo] + (g - g0s[lo])) * 8 + (ci & 7));
                    const unsigned vb = vcode[rs + j - vd_base];
                    vv[q] = vb == 255u && diag_idx >= 0 ? (unsigned)diag_idx : vb;
                }
            }
            uint32_t *dst = codes + ((size_t)(c0 + c) * 64 + lane) * words;
            if (fmt == 2) { // slot : 13 | index : 3
                dst[0] = (cc[0] << 3 | vv[0]) | (cc[1] << 3 | vv[1]) << 16;
                dst[1] = (cc[2] << 3 | vv[2]) | (cc[3] << 3 | vv[3]) << 16;
            } else { // byte offsets of the slots, then the value bytes (fmt 1: 8 * index)
                const unsigned sh = fmt == 1 ? 3 : 0;
                dst[0] = (cc[0] * 8u) | (cc[1] * 8u) << 16;
                dst[1] = (cc[2] * 8u) | (cc[3] * 8u) << 16;
                dst[2] = (vv[0] << sh) | (vv[1] << sh) << 8 | (vv[2] << sh) << 16 | (vv[3] << sh) << 24;
            }
        }
    }
}

// fmt 3, step 1: the distinct (column - row, value code) pairs of the matrix, one list per wave (a workgroup is one
// wave): lists[w * 257] = count (more than 254: *overflow is raised and every wave stops), then the keys.
template <typename RP>
__global__ __launch_bounds__(64) void sw_pairs_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                      const uint8_t *__restrict__ vcode, int64_t vd_base, int64_t n_rows, int64_t row0,
                                                      unsigned long long *__restrict__ lists, int *overflow) {
    __shared__ unsigned long long list[256];
    const int lane = threadIdx.x;
    int n = 0;
    for (int64_t rb = (int64_t)blockIdx.x * 64; rb < n_rows && n <= kSwMaxPairs; rb += (int64_t)gridDim.x * 64) {
        if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { n = kSwMaxPairs + 1; break; }
        const int64_t r = rb + lane;
        int64_t k = 0, k1 = 0;
        if (r < n_rows) { k = (int64_t)row_ptr[r]; k1 = (int64_t)row_ptr[r + 1]; }
        while (__ballot(k < k1) && n <= kSwMaxPairs) {
            bool found = true;
            unsigned long long v = 0;
            if (k < k1) {
                v = sw_pair_key(col[k] - (int32_t)(r + row0), vcode[k - vd_base]);
                found = false;
                ++k;
            }
            for (int j = 0; j < n && __ballot(!found); ++j) found |= list[j] == v;
            while (const unsigned long long open = __ballot(!found)) {
                const int leader = (int)__builtin_ctzll(open);
                const unsigned long long lv = __shfl(v, leader);
                if (n == kSwMaxPairs) { n = kSwMaxPairs + 1; break; }
                if (lane == 0) list[n] = lv;
                ++n;
                found |= v == lv;
            }
            __syncthreads();
        }
    }
    if (n > kSwMaxPairs && lane == 0) atomicExch(overflow, 1);
    if (lane == 0) lists[(size_t)blockIdx.x * 257] = (unsigned long long)n;
    for (int j = lane; j < n && j < 256; j += 64) lists[(size_t)blockIdx.x * 257 + 1 + j] = list[j];
}

// fmt 3, step 2: one byte per non-zero = index of its pair; the block's bases.  A pair whose columns do not run through
// the window in step with the rows (slot != base + row for some row of the block) raises status[2]: another format then.
template <typename RP>
__global__ __launch_bounds__(256) void sw_fill_pairs_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                            const uint8_t *__restrict__ vcode, int64_t vd_base, int64_t n_rows, int64_t row0,
                                                            int R, const int32_t *__restrict__ hdr, const int64_t *__restrict__ slice_chunk0,
                                                            uint32_t *__restrict__ codes, const unsigned long long *__restrict__ pair_key,
                                                            int n_pairs, int pair_stride, int16_t *__restrict__ blk_base, int *status) {
    __shared__ int g0s[kSwRuns], rk[kSwRuns];
    __shared__ int nr_s;
    __shared__ unsigned long long keys[256];
    __shared__ int base[256];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    keys[tid] = pair_key[tid];
    base[tid] = INT32_MIN;
    if (tid < kSwRuns) {
        g0s[tid] = hdr[(size_t)b * 64 + tid];
        const int w2 = hdr[(size_t)b * 64 + 32 + tid];
        rk[tid] = w2 & 0xffff;
        const unsigned long long m = __ballot((w2 >> 16) != 0);
        if (tid == 0) nr_s = __popcll(m);
    }
    __syncthreads();
    const int nr = nr_s;
    bool bad = false;
    for (int rr = 0; rr < R; ++rr) {
        const int64_t slice = ((int64_t)b * 4 + wv) * R + rr;
        const int64_t r = slice * 64 + lane;
        const int r_in_block = (wv * R + rr) * 64 + lane;
        int64_t rs = 0;
        int len = 0;
        if (r < n_rows) {
            rs = (int64_t)row_ptr[r];
            len = (int)((int64_t)row_ptr[r + 1] - rs);
        }
        const int64_t c0 = slice_chunk0[slice];
        const int nch = (int)(slice_chunk0[slice + 1] - c0);
        for (int c = 0; c < nch; ++c) {
            unsigned word = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 4 * c + q;
                unsigned e = (unsigned)n_pairs; // padding
                if (j < len) {
                    const int ci = col[rs + j];
                    const unsigned long long key = sw_pair_key(ci - (int32_t)(r + row0), vcode[rs + j - vd_base]);
                    int lo = 0, hi = n_pairs - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (keys[mid] < key) lo = mid + 1; else hi = mid;
                    }
                    e = (unsigned)lo;
                    const int g = ci >> 3;
                    int rl = 0, rh = nr - 1;
                    while (rl < rh) {
                        const int mid = (rl + rh + 1) >> 1;
                        if (g0s[mid] <= g) rl = mid; else rh = mid - 1;
                    }
                    const int delta = 2 + (rk[rl] + (g - g0s[rl])) * 8 + (ci & 7) - r_in_block;
                    const int old = atomicCAS(&base[lo], INT32_MIN, delta);
                    bad |= old != INT32_MIN && old != delta;
                }
                word |= e << (8 * q);
            }
            codes[(size_t)(c0 + c) * 64 + lane] = word;
        }
    }
    if (bad) atomicExch(&status[2], 1);
    __syncthreads();
    if (tid < pair_stride) blk_base[(size_t)b * pair_stride + tid] = base[tid] == INT32_MIN ? (int16_t)0 : (int16_t)base[tid];
}

// fmt 4: 32 bits per row -- bit e set: the row has pair e -- and the block's bases as in fmt 3.  Applies where the matrix has at
// most 32 pairs and every row's entries run through the (ascending) pair list in ascending order, which a row with ascending
// columns does: then "for e ascending: if bit e: acc += value(e) * x(e)" IS the row's CRS-ordered sum.  status[2] otherwise.
// Pair e is bit 31 - e: the kernel shifts the word left a bit per pair and takes the carry.
template <typename RP>
__global__ __launch_bounds__(256) void sw_fill_masks_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                            const uint8_t *__restrict__ vcode, int64_t vd_base, int64_t n_rows, int64_t row0,
                                                            int R, const int32_t *__restrict__ hdr, uint32_t *__restrict__ masks,
                                                            const unsigned long long *__restrict__ pair_key, int n_pairs, int pair_stride,
                                                            int16_t *__restrict__ blk_base, int *status) {
    __shared__ int g0s[kSwRuns], rk[kSwRuns];
    __shared__ int nr_s;
    __shared__ unsigned long long keys[32];
    __shared__ int base[32];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 32) {
        keys[tid] = pair_key[tid];
        base[tid] = INT32_MIN;
    }
    if (tid < kSwRuns) {
        g0s[tid] = hdr[(size_t)b * 64 + tid];
        const int w2 = hdr[(size_t)b * 64 + 32 + tid];
        rk[tid] = w2 & 0xffff;
        const unsigned long long m = __ballot((w2 >> 16) != 0);
        if (tid == 0) nr_s = __popcll(m);
    }
    __syncthreads();
    const int nr = nr_s;
    bool bad = false;
    for (int rr = 0; rr < R; ++rr) {
        const int64_t slice = ((int64_t)b * 4 + wv) * R + rr;
        const int64_t r = slice * 64 + lane;
        const int r_in_block = (wv * R + rr) * 64 + lane;
        uint32_t mask = 0;
        if (r < n_rows) {
            const int64_t rs = (int64_t)row_ptr[r], re = (int64_t)row_ptr[r + 1];
            int prev = -1;
            for (int64_t k = rs; k < re; ++k) {
                const int ci = col[k];
                const unsigned long long key = sw_pair_key(ci - (int32_t)(r + row0), vcode[k - vd_base]);
                int lo = 0, hi = n_pairs - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (keys[mid] < key) lo = mid + 1; else hi = mid;
                }
                bad |= lo <= prev; // not in the list's order (or the same pair twice): the mask cannot say it
                prev = lo;
                mask |= 1u << lo;
                const int g = ci >> 3;
                int rl = 0, rh = nr - 1;
                while (rl < rh) {
                    const int mid = (rl + rh + 1) >> 1;
                    if (g0s[mid] <= g) rl = mid; else rh = mid - 1;
                }
                const int delta = 2 + (rk[rl] + (g - g0s[rl])) * 8 + (ci & 7) - r_in_block;
                const int old = atomicCAS(&base[lo], INT32_MIN, delta);
                bad |= old != INT32_MIN && old != delta;
            }
        }
        masks[(size_t)slice * 64 + lane] = __brev(mask);
    }
    if (bad) atomicExch(&status[2], 1);
    __syncthreads();
    // (fmt 4 keeps the bases as LDS byte offsets of 32 bits -- scalar loads have no 16-bit form --, 32 per block in the space of 64 int16)
    if (tid < 32) reinterpret_cast<int32_t *>(blk_base)[(size_t)b * 32 + tid] = base[tid] == INT32_MIN ? 0 : 8 * base[tid];
}

template <int FMT> struct sw_chunk { uint32_t a, b, c; };
template <> struct sw_chunk<2> { uint32_t a, b; };
template <> struct sw_chunk<3> { uint32_t a; };

template <bool DIAG, int FMT, int R>
struct SwLayout {
    // (fmt 3 sizes its table by the matrix' pairs: its offsets are run-time values, sw_layout3)
    static constexpr int kDictBytes = FMT == 2 ? 64 : (FMT == 1 ? 256 : 2048);
    static constexpr int kDiagOff = kDictBytes;
    static constexpr int kWinOff = kDictBytes + (DIAG ? 2048 * R : 0);
    static constexpr unsigned kDiagCode = FMT == 2 ? 7u : (FMT == 1 ? 248u : 255u); // the value code that stands for the row's diagonal
};

// four non-zeros of every row of the wave: acc += table[value code] * window[column code], in entry order
// fmt 3.  LDS: table entries of 16 bytes {value; LDS address of the pair's column for the block's first row; mask}, the
// per-row diagonal values (DIAG), one slot of -0.0, the window.  x address = entry address + (8 * row in block & mask): the
// padding entry points at the -0.0 slot with mask 0, whatever the lane.
struct SwLayout3 { int tab_bytes, diag_off, pad_off, win_off; };
__host__ __device__ inline SwLayout3 sw_layout3(int n_pairs, bool diag, int R) {
    SwLayout3 l;
    l.tab_bytes = (16 * (n_pairs + 1) + 255) & ~255;
    l.diag_off = l.tab_bytes;
    l.pad_off = l.diag_off + (diag ? 2048 * R : 0);
    l.win_off = l.pad_off; // the window's own first slot is the -0.0 one
    return l;
}
template <bool DIAG, int FMT, int R>
__device__ __forceinline__ void sw_chunk_fma(const unsigned char *lds, const sw_chunk<FMT> &cd, unsigned diag_rel, double &acc) {
#pragma clang fp contract(off)
    using L = SwLayout<DIAG, FMT, R>;
    const unsigned code[4] = {cd.a & 0xffffu, cd.a >> 16, cd.b & 0xffffu, cd.b >> 16};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned xa, vc, va;
        if constexpr (FMT == 2) {
            xa = code[q] & 0xfff8u;
            vc = code[q] & 7u;
            va = vc << 3;
        } else {
            xa = code[q];
            vc = (cd.c >> (8 * q)) & 0xffu;
            va = FMT == 1 ? vc : vc << 3;
        }
        if (DIAG) va = vc == L::kDiagCode ? diag_rel : va;
        const double xv = *reinterpret_cast<const double *>(lds + L::kWinOff + xa);
        const double v = *reinterpret_cast<const double *>(lds + va);
        const double pr = v * xv;
        acc = acc + pr;
    }
}

// fmt 3, Q chunks of a slice, software-pipelined by hand: a non-zero costs two DEPENDENT LDS round trips (its table entry,
// then the x slot the entry points at), and with the chunks taken one after the other a wave has only one of them in flight.
// Here the table entries of chunk q + 1 are read before chunk q's x slots are: one round trip per chunk instead of two.
struct sw_ent { double v; unsigned xo, mask; };
template <bool DIAG>
__device__ __forceinline__ void sw_read_ents(const unsigned char *lds, unsigned word, unsigned diag_rel, unsigned diag_code, sw_ent (&e)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned ea = ((word >> (8 * q)) & 0xffu) << 4;
        const uint2 xm = *reinterpret_cast<const uint2 *>(lds + ea + 8);
        unsigned va = ea;
        if (DIAG) va = ea == diag_code ? diag_rel : ea;
        e[q].v = *reinterpret_cast<const double *>(lds + va);
        e[q].xo = xm.x;
        e[q].mask = xm.y;
    }
}
template <int Q, bool DIAG, int R>
__device__ __forceinline__ void sw_consume3(const unsigned char *lds, const sw_chunk<3> (&cd)[8], unsigned diag_rel, unsigned diag_code,
                                            unsigned row_off, double &acc) {
#pragma clang fp contract(off)
    sw_ent cur[4], nxt[4];
    sw_read_ents<DIAG>(lds, cd[0].a, diag_rel, diag_code, cur);
#pragma unroll
    for (int c = 0; c < Q; ++c) {
        double xv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[q] = *reinterpret_cast<const double *>(lds + (cur[q].xo + (row_off & cur[q].mask)));
        if (c + 1 < Q) sw_read_ents<DIAG>(lds, cd[c + 1].a, diag_rel, diag_code, nxt);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double pr = cur[q].v * xv[q];
            acc = acc + pr;
        }
        if (c + 1 < Q) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        }
    }
}

template <int Q, bool DIAG, int FMT, int R>
__device__ __forceinline__ void sw_consume(const unsigned char *lds, const sw_chunk<FMT> (&cd)[8], unsigned diag_rel, unsigned diag_code,
                                           unsigned row_off, double &acc) {
    if constexpr (FMT == 3) {
        sw_consume3<Q, DIAG, R>(lds, cd, diag_rel, diag_code, row_off, acc);
    } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) sw_chunk_fma<DIAG, FMT, R>(lds, cd[q], diag_rel, acc);
    }
}

// MODE 0: y = A x.  MODE 1: also partials[slice] = sum over the slice's rows of y[r] w[r] (CG's (Ap, p)).
// NCH > 0: every slice of the matrix has exactly NCH chunks (sw->uniform_chunks; fmt 3 only): the chunk walk is straight-line
// code, which is what lets the hand-pipelined LDS reads of sw_consume3 overlap (a walk that can end after any chunk is a chain
// of branches, and the compiler keeps each chunk's reads behind the branch before it).
template <int MODE, bool DIAG, int FMT, int R, int NCH = 0>
__global__ __launch_bounds__(256) void spmv_sellwin_kernel(
    const double *x, double *__restrict__ y, int64_t n_rows, int64_t n_cols, int n_blocks, int remap_arg, const double *w,
    double *__restrict__ partials, const int *stop, const int32_t *__restrict__ hdr, const int64_t *__restrict__ slice_chunk0,
    const uint32_t *__restrict__ codes, const double *__restrict__ dict_g, const double *__restrict__ vdiag, int x_flags,
    const int16_t *__restrict__ blk_base, int pair_stride, int n_pairs, int diag_pair, const int32_t *__restrict__ own_rank, long long *dbg) {
    using L = SwLayout<DIAG, FMT, R>;
    using Chunk = sw_chunk<FMT>;
    const bool x_al16 = (x_flags & 1) != 0, nt_codes = (x_flags & 2) != 0; // x 16-byte aligned; the code stream read non-temporally
    if (stop && stop[1]) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int b = remap_arg > 0 ? xcd_remap(blockIdx.x, remap_arg)
                                : (remap_arg < -1 ? xcd_group_remap(blockIdx.x, -remap_arg) : (int)blockIdx.x);
    if (b >= n_blocks) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long t_start = dbg ? (long long)__builtin_readcyclecounter() : 0; // diagnostic stamps (BIS_SELLWIN_DEBUG): shader cycles
    const int hw = hdr[(size_t)b * 64 + lane];
    // (pulling the header of a block 512 or 2048 places ahead into the caches was measured: no change -- the ~4000 cycles until
    // the header arrives are queueing in the vector memory path under load, not a cache miss)
    const int64_t slice0 = ((int64_t)b * 4 + wv) * R; // this wave's R consecutive slices
    int64_t c0[R];
    int nch[R];
    Chunk cd[R][8];
    double wr[R], dval[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if constexpr (NCH > 0) {
            c0[r] = (slice0 + r) * NCH;
            nch[r] = NCH;
        } else {
            c0[r] = slice_chunk0[slice0 + r];
            nch[r] = (int)(slice_chunk0[slice0 + r + 1] - c0[r]);
        }
        const Chunk *cp = reinterpret_cast<const Chunk *>(codes) + (size_t)c0[r] * 64 + lane;
        const int last = max(nch[r] - 1, 0); // (the stream ends with one spare chunk: an empty last slice reads it)
        if (FMT == 3 && nt_codes) { // (read once per product: kept out of the caches the vectors of the iteration live in)
#pragma unroll
            for (int q = 0; q < 8; ++q) cd[r][q].a = __builtin_nontemporal_load(&cp[(size_t)min(q, last) * 64].a);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) cd[r][q] = cp[(size_t)min(q, last) * 64];
        }
        const int64_t row = (slice0 + r) * 64 + lane;
        wr[r] = 0.0;
        if (MODE == 1 && !own_rank && row < n_rows) wr[r] = w[row];
        if (DIAG) dval[r] = row < n_rows ? vdiag[row] : 0.0;
    }
    // own_rank != nullptr: w is x at the rows' own columns (CG: w = x = p) -- where the block's window holds them side by side
    // the dot's operand comes from LDS behind the barrier instead of a second global read of p
    int own = -1;
    if (MODE == 1 && own_rank) {
        own = own_rank[b];
        if (own < 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) { const int64_t row = (slice0 + r) * 64 + lane; if (row < n_rows) wr[r] = w[row]; }
        }
    }
    // tables
    const SwLayout3 l3 = sw_layout3(n_pairs, DIAG, R);
    const int win_off = FMT == 3 ? l3.win_off : L::kWinOff, diag_off = FMT == 3 ? l3.diag_off : L::kDiagOff;
    if constexpr (FMT == 3) { // entry e: {value of pair e; LDS address of its column for the block's first row; mask}; the padding: 1.0 at the -0.0 slot
        if (tid <= n_pairs) {
            int xo = l3.pad_off, mask = 0;
            if (tid < n_pairs) { xo = l3.win_off + 8 * (int)blk_base[(size_t)b * pair_stride + tid]; mask = -1; }
            *reinterpret_cast<double *>(lds + 16 * tid) = dict_g[tid];
            *reinterpret_cast<int2 *>(lds + 16 * tid + 8) = make_int2(xo, mask);
        }
    } else if (tid < L::kDictBytes / 8) reinterpret_cast<double *>(lds)[tid] = dict_g[tid];
    if (DIAG) {
#pragma unroll
        for (int r = 0; r < R; ++r) reinterpret_cast<double *>(lds + diag_off)[(wv * R + r) * 64 + lane] = dval[r];
    }
    if (tid == 0) *reinterpret_cast<double *>(lds + win_off) = -0.0;
    // window: the runs of 8-column granules; a run has at most 256 pieces of 16 bytes, a wave takes 64 of them and
    // the hardware writes them to LDS behind the wave-uniform base (no register staging, nothing waited for here)
    {
        const int n_runs = __popcll(__ballot(lane >= 32 && (hw >> 16) != 0));
        if (dbg && tid == 0) dbg[(size_t)b * 4 + 1] = (long long)__builtin_readcyclecounter() - t_start; // header arrived
        unsigned char *win = lds + win_off + 16;
        for (int k = 0; k < n_runs; ++k) {
            const int g0 = __builtin_amdgcn_readlane(hw, k);
            const int w2 = __builtin_amdgcn_readlane(hw, 32 + k);
            const int rank = w2 & 0xffff, n_pieces = (w2 >> 16) * 4;
            const int j = (wv + k) & 3;
            const int p = j * 64 + lane;
            const int64_t c = (int64_t)g0 * 8 + 2 * p;
            unsigned char *dst = win + (size_t)rank * 64 + (size_t)j * 1024;
            if (p < n_pieces) {
                if (x_al16 && c + 1 < n_cols) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + c),
                                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                } else {
                    double2 v;
                    v.x = c < n_cols ? x[c] : 0.0;
                    v.y = c + 1 < n_cols ? x[c + 1] : 0.0;
                    *reinterpret_cast<double2 *>(dst + lane * 16) = v;
                }
            }
        }
    }
    __syncthreads();
    if (dbg && tid == 0) dbg[(size_t)b * 4 + 2] = (long long)__builtin_readcyclecounter() - t_start; // window and codes arrived, barrier passed
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned diag_rel = (unsigned)(diag_off + ((wv * R + r) * 64 + lane) * 8);
        const unsigned row_off = (unsigned)(((wv * R + r) * 64 + lane) * 8), diag_code = (unsigned)diag_pair << 4;
        const Chunk *cp = reinterpret_cast<const Chunk *>(codes) + (size_t)c0[r] * 64 + lane;
        double acc = 0.0;
        if constexpr (NCH > 0) sw_consume<NCH, DIAG, FMT, R>(lds, cd[r], diag_rel, diag_code, row_off, acc);
        else
        for (int cb = 0; cb < nch[r]; cb += 8) {
            const int rem = nch[r] - cb;
            if (cb > 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) cd[r][q] = cp[(size_t)(cb + min(q, rem - 1)) * 64];
            }
            switch (rem) {
            case 1: sw_consume<1, DIAG, FMT, R>(lds, cd[r], diag_rel, diag_code, row_off, acc); break;
            case 2: sw_consume<2, DIAG, FMT, R>(lds, cd[r], diag_rel, diag_code, row_off, acc); break;
            case 3: sw_consume<3, DIAG, FMT, R>(lds, cd[r], diag_rel, diag_code, row_off, acc); break;
            case 4: sw_consume<4, DIAG, FMT, R>(lds, cd[r], diag_rel, diag_code, row_off, acc); break;
            case 5: sw_consume<5, DIAG, FMT, R>(lds, cd[r], diag_rel, diag_code, row_off, acc); break;
            case 6: sw_consume<6, DIAG, FMT, R>(lds, cd[r], diag_rel, diag_code, row_off, acc); break;
            case 7: sw_consume<7, DIAG, FMT, R>(lds, cd[r], diag_rel, diag_code, row_off, acc); break;
            default: sw_consume<8, DIAG, FMT, R>(lds, cd[r], diag_rel, diag_code, row_off, acc); break;
            }
        }
        const int64_t row = (slice0 + r) * 64 + lane;
        if (row < n_rows) y[row] = acc;
        if (MODE == 1) {
            if (own >= 0) wr[r] = *reinterpret_cast<const double *>(lds + win_off + 16 + ((size_t)own * 8 + (wv * R + r) * 64 + lane) * 8);
            const double t = wave_sum(row < n_rows ? acc * wr[r] : 0.0);
            if (lane == 0) partials[slice0 + r] = t;
        }
    }
    if (dbg && tid == 0) { dbg[(size_t)b * 4 + 3] = (long long)__builtin_readcyclecounter() - t_start; dbg[(size_t)b * 4] = t_start; }
}

// fmt 4: y = A x from 32 bits per row.  The (at most 32) pairs of the matrix are the same for every row; a wave-uniform step per
// pair has the pair's value and the LDS address of its column for the block's first row in scalar registers, and the rows that
// have the pair read their x entry from the window and add value * x -- in pair order, which is the rows' CRS order.  "The rows
// that have the pair" is the hardware's execution mask: v_add_co_u32 mask, mask, mask shifts a row's word left and delivers the
// bit that falls out -- the next pair's -- for all 64 rows as a scalar register pair, ONE vector instruction per row and pair.
// Per non-zero: nothing streamed (4 bytes per ROW), ONE LDS read of 8 bytes, that add, a multiply and an add; the byte codes of
// fmt 3 cost two dependent LDS reads (24 bytes of LDS traffic per non-zero: at HPCG-256 75 % of that kernel's time on the LDS
// pipe alone) and some ten vector instructions.  Four pairs a step and all the wave's R slices at once: 4 R reads in flight,
// then 4 R multiply-adds, each under its rows' mask -- hand-written: the compiler turns "if (bit) acc += v * x" into a branch and
// a full LDS wait per pair and row.  exec is all ones on entry (no divergent control flow around the steps) and on exit.
#define SM_BIT(r, q) "v_add_co_u32 %[mk" #r "], %[m" #r #q "], %[mk" #r "], %[mk" #r "]\n\t"
#define SM_BIT_ROW(r) SM_BIT(r, 0) SM_BIT(r, 1) SM_BIT(r, 2) SM_BIT(r, 3)
#define SM_RD(r, q, off) "s_mov_b64 exec, %[m" #r #q "]\n\tds_read_b64 %[x" #r #q "], %[a" #q "] offset:" #off "\n\t"
#define SM_RD_ROW(r, off) SM_RD(r, 0, off) SM_RD(r, 1, off) SM_RD(r, 2, off) SM_RD(r, 3, off)
#define SM_MA(r, q) "s_mov_b64 exec, %[m" #r #q "]\n\tv_mul_f64 %[x" #r #q "], %[v" #q "], %[x" #r #q "]\n\tv_add_f64 %[acc" #r "], %[acc" #r "], %[x" #r #q "]\n\t"
#define SM_MA_ROW(r) SM_MA(r, 0) SM_MA(r, 1) SM_MA(r, 2) SM_MA(r, 3)
#define SM_ADDR "v_add_u32 %[a0], %[xo0], %[roff]\n\tv_add_u32 %[a1], %[xo1], %[roff]\n\tv_add_u32 %[a2], %[xo2], %[roff]\n\tv_add_u32 %[a3], %[xo3], %[roff]\n\t"
#define SM_WAIT "s_waitcnt lgkmcnt(0)\n\t"
#define SM_END "s_mov_b64 exec, -1"
#define SM_OUT_ROW(r) [acc##r] "+v"(acc[r]), [mk##r] "+v"(mk[r]), [x##r##0] "=&v"(x[r][0]), [x##r##1] "=&v"(x[r][1]), [x##r##2] "=&v"(x[r][2]), [x##r##3] "=&v"(x[r][3]), \
                      [m##r##0] "=&s"(m[r][0]), [m##r##1] "=&s"(m[r][1]), [m##r##2] "=&s"(m[r][2]), [m##r##3] "=&s"(m[r][3])
#define SM_OUT_A [a0] "=&v"(a[0]), [a1] "=&v"(a[1]), [a2] "=&v"(a[2]), [a3] "=&v"(a[3])
#define SM_IN [roff] "v"(roff), [xo0] "s"(xo[0]), [xo1] "s"(xo[1]), [xo2] "s"(xo[2]), [xo3] "s"(xo[3]), [v0] "s"(sv[0]), [v1] "s"(sv[1]), [v2] "s"(sv[2]), [v3] "s"(sv[3])
// PRECONDITION: exec is all ones on entry (the block rewrites exec and leaves it all ones: SM_END).  Every call site sits in
// wave-uniform control flow -- the pair loop's bounds and the block's row count are scalar -- and a kernel is entered with a full
// execution mask (256-thread workgroups).  hipcc rejects exec on a clobber list ("reserved register"), so the rule is this
// comment: never call sm_step under a lane-dependent branch.
template <int R>
__device__ __forceinline__ void sm_step(double (&acc)[R], uint32_t (&mk)[R], unsigned roff, const unsigned (&xo)[4], const double (&sv)[4]) {
    unsigned a[4];
    double x[R][4];
    unsigned long long m[R][4];
    if constexpr (R == 1)
        asm volatile(SM_ADDR SM_BIT_ROW(0) SM_RD_ROW(0, 0) SM_WAIT SM_MA_ROW(0) SM_END : SM_OUT_ROW(0), SM_OUT_A : SM_IN);
    else if constexpr (R == 2)
        asm volatile(SM_ADDR SM_BIT_ROW(0) SM_BIT_ROW(1) SM_RD_ROW(0, 0) SM_RD_ROW(1, 512) SM_WAIT SM_MA_ROW(0) SM_MA_ROW(1) SM_END
                     : SM_OUT_ROW(0), SM_OUT_ROW(1), SM_OUT_A : SM_IN);
    else
        asm volatile(SM_ADDR SM_BIT_ROW(0) SM_BIT_ROW(1) SM_BIT_ROW(2) SM_BIT_ROW(3) SM_RD_ROW(0, 0) SM_RD_ROW(1, 512) SM_RD_ROW(2, 1024)
                     SM_RD_ROW(3, 1536) SM_WAIT SM_MA_ROW(0) SM_MA_ROW(1) SM_MA_ROW(2) SM_MA_ROW(3) SM_END
                     : SM_OUT_ROW(0), SM_OUT_ROW(1), SM_OUT_ROW(2), SM_OUT_ROW(3), SM_OUT_A : SM_IN);
}

// NP > 0: the number of pairs, known at compile time.  Four waves of R slices each: a block of 256 R rows.  LDS: 16 spare bytes,
// then the window.
// (Measured and removed: the pairs' constants and / or transposed per-slice masks through scalar loads instead of v_readlane /
// v_add_co -- 0.150 against 0.135 ms at HPCG-256, the scalar loads' latency is not hidden; eight waves of two slices on the
// 1024-row block -- 0.131 against 0.124; a workgroup walking 2 / 4 / 8 consecutive blocks with the next block's header, masks and
// bases loaded under the current one -- 0.115 / 0.123 / 0.125 against 0.118: the header trip disappears from the stamps and the
// block's life stays the same.  rocprofv3 (profiles/r04_e_spmv_pmc_sellmask.txt): 11 instructions per non-zero and wave -- 5.4
// vector, 4.8 scalar (the execution-mask moves), 1 LDS -- at 3.4 waves per SIMD: instruction issue, not a memory pipe, paces it.)
template <int MODE, int R, int NP>
__global__ __launch_bounds__(256) void spmv_sellmask_kernel(
    const double *x, double *__restrict__ y, int64_t n_rows, int64_t n_cols, int n_blocks, int remap_arg, const double *w,
    double *__restrict__ partials, const int *stop, const int32_t *__restrict__ hdr, const uint32_t *__restrict__ masks,
    const double *__restrict__ dict_g, int x_flags, const int32_t *__restrict__ blk_base32, int n_pairs_arg,
    const int32_t *__restrict__ own_rank, long long *dbg) {
    const bool x_al16 = (x_flags & 1) != 0, nt_codes = (x_flags & 2) != 0;
    if (stop && stop[1]) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int b = remap_arg > 0 ? xcd_remap(blockIdx.x, remap_arg)
                                : (remap_arg < -1 ? xcd_group_remap(blockIdx.x, -remap_arg) : (int)blockIdx.x);
    if (b >= n_blocks) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_pairs = NP > 0 ? NP : n_pairs_arg;
    const long long t_start = dbg ? (long long)__builtin_readcyclecounter() : 0; // diagnostic stamps (BIS_SELLWIN_DEBUG): shader cycles
    const int hw = hdr[(size_t)b * 64 + lane];
    const int64_t slice0 = ((int64_t)b * 4 + wv) * R;
    // pair `lane`: its value and where its column for the block's first row lies in LDS
    double pv = 0.0;
    int pxo = 0;
    if (lane < n_pairs) {
        pv = dict_g[lane];
        pxo = blk_base32[(size_t)b * 32 + lane];
    }
    uint32_t mk[R];
    double wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = (slice0 + r) * 64 + lane;
        mk[r] = nt_codes ? __builtin_nontemporal_load(&masks[row]) : masks[row]; // (the array is padded to whole blocks)
        wr[r] = 0.0;
        if (MODE == 1 && !own_rank && row < n_rows) wr[r] = w[row];
    }
    int own = -1;
    if (MODE == 1 && own_rank) {
        own = own_rank[b];
        if (own < 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) { const int64_t row = (slice0 + r) * 64 + lane; if (row < n_rows) wr[r] = w[row]; }
        }
    }
    { // the window: as in spmv_sellwin_kernel
        const int n_runs = __popcll(__ballot(lane >= 32 && (hw >> 16) != 0));
        if (dbg && tid == 0) dbg[(size_t)b * 4 + 1] = (long long)__builtin_readcyclecounter() - t_start; // header arrived
        unsigned char *win = lds + 16;
        for (int k = 0; k < n_runs; ++k) {
            const int g0 = __builtin_amdgcn_readlane(hw, k);
            const int w2 = __builtin_amdgcn_readlane(hw, 32 + k);
            const int rank = w2 & 0xffff, n_pieces = (w2 >> 16) * 4;
            const int j = (wv + k) & 3;
            const int p = j * 64 + lane;
            const int64_t c = (int64_t)g0 * 8 + 2 * p;
            unsigned char *dst = win + (size_t)rank * 64 + (size_t)j * 1024;
            if (p < n_pieces) {
                if (x_al16 && c + 1 < n_cols) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + c),
                                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                } else {
                    double2 v;
                    v.x = c < n_cols ? x[c] : 0.0;
                    v.y = c + 1 < n_cols ? x[c + 1] : 0.0;
                    *reinterpret_cast<double2 *>(dst + lane * 16) = v;
                }
            }
        }
    }
    __syncthreads();
    if (dbg && tid == 0) dbg[(size_t)b * 4 + 2] = (long long)__builtin_readcyclecounter() - t_start; // window arrived, barrier passed
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    const unsigned row_off0 = (unsigned)((wv * R * 64 + lane) * 8); // (the wave's slice r: + 512 r)
    const unsigned pv_lo = (unsigned)(unsigned long long)__double_as_longlong(pv), pv_hi = (unsigned)((unsigned long long)__double_as_longlong(pv) >> 32);
    auto step = [&](int e0) { // pairs e0 .. e0 + 3 (no row has the ones past the last)
        unsigned xo[4];
        double sv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = min(e0 + q, 31);
            xo[q] = (unsigned)__builtin_amdgcn_readlane(pxo, e);
            sv[q] = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)pv_hi, e) << 32) |
                                                     (unsigned)__builtin_amdgcn_readlane((int)pv_lo, e)));
        }
        sm_step<R>(acc, mk, row_off0, xo, sv);
    };
    if constexpr (NP > 0) {
#pragma unroll
        for (int e0 = 0; e0 < NP; e0 += 4) step(e0);
    } else {
        for (int e0 = 0; e0 < n_pairs; e0 += 4) step(e0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = (slice0 + r) * 64 + lane;
        if (row < n_rows) y[row] = acc[r];
        if (MODE == 1) {
            if (own >= 0) wr[r] = *reinterpret_cast<const double *>(lds + 16 + ((size_t)own * 8 + (wv * R + r) * 64 + lane) * 8);
            const double t = wave_sum(row < n_rows ? acc[r] * wr[r] : 0.0);
            if (lane == 0) partials[slice0 + r] = t;
        }
    }
    if (dbg && tid == 0) { dbg[(size_t)b * 4 + 3] = (long long)__builtin_readcyclecounter() - t_start; dbg[(size_t)b * 4] = t_start; }
}

// (Measured and removed: a "chained" variant of the fmt 3 kernel in which one workgroup walks 2 / 4 / 8 consecutive blocks and
// loads the next block's header, codes and pair bases while the current block computes, so that only a workgroup's first
// block pays for the header trip: HPCG-256 in the CG loop 0.218 / 0.224 / 0.235 ms against 0.205 ms with one block per
// workgroup -- 92-100 registers instead of 54 cost a resident workgroup per CU, and the header trip it hides overlaps with
// other workgroups' work anyway.)
// (Also measured and removed: a double-buffered, software-pipelined variant -- a workgroup walks 4 / 8 / 16 consecutive blocks
// with TWO window buffers in LDS; while block i computes, the window loads of block i + 1 (LDS-DMA issued as an asm statement
// so that hipcc does not drain it in front of the computation's first LDS read; __builtin_amdgcn_s_waitcnt at the top of each
// step so that it knows the computing set's registers are ready) and the header / codes of block i + 2 are in flight; three
// register sets rotate by a three-fold unroll, no copies.  The ISA shows exactly the intended overlap and the results are
// bit-identical, but HPCG-256 in the CG loop takes 0.271 / 0.275 / 0.278 ms against 0.206: twice the LDS per workgroup
// leaves three workgroups = three waves per SIMD, and the computation -- two dependent LDS round trips per four non-zeros
// -- needs the six waves per SIMD of the single-buffer kernel to cover its own LDS latency.)
bool sw_enabled() { return bis_opts().spmv_sellwin != 0; }

} // namespace

void bis_spmv_sellwin_drop(bis_mat *A) {
    if (A->sw) {
        hipFree(A->sw->hdr); hipFree(A->sw->slice_chunk0); hipFree(A->sw->own_rank); hipFree(A->sw->codes); hipFree(A->sw->dict); hipFree(A->sw->blk_base); hipFree(A->sw->pair_key);
        delete A->sw;
        A->sw = nullptr;
    }
    A->sw_state = 0;
}

#define SW_CHECK(call)                                                         \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) {                                                \
            (void)hipGetLastError();                                           \
            hipFree(slice_chunks); hipFree(tmp);                               \
            bis_spmv_sellwin_drop(A);                                          \
            A->sw_state = -1;                                                  \
            if (e_ == hipErrorOutOfMemory) return BIS_OK; /* the gather kernels stay */ \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);      \
            return BIS_ERR_HIP;                                                \
        }                                                                      \
    } while (0)

// The distinct (column - row, value code) pairs of the matrix, sorted; `known` says the list has been made, `fits` that there are
// 1 .. kSwMaxPairs of them.  Made once per build (the list depends on the matrix only, not on the block size of the plan).
struct SwPairs { bool known = false, fits = false; std::vector<unsigned long long> all; };

static bis_status sw_collect_pairs(bis_ctx *ctx, bis_mat *A, SwPairs &P) {
    P.known = true; P.fits = false; P.all.clear();
    int32_t *slice_chunks = nullptr; // (SW_CHECK's clean-up names)
    void *tmp = nullptr;
    const int n_waves = (int)std::min<int64_t>((A->n_rows + 63) / 64, (int64_t)ctx->n_cus * 8);
    unsigned long long *lists = nullptr;
    int *status = (int *)ctx->counters + 52;
    SW_CHECK(hipMalloc(&lists, sizeof(unsigned long long) * 257 * (size_t)n_waves));
    tmp = lists;
    SW_CHECK(hipMemsetAsync(status + 2, 0, 2 * sizeof(int), ctx->stream));
    if (A->rp64) hipLaunchKernelGGL(sw_pairs_kernel<int64_t>, dim3(n_waves), dim3(64), 0, ctx->stream, (const int64_t *)A->row_ptr, A->col, A->vcode, A->vd_base, A->n_rows, A->view_row0, lists, status + 3);
    else hipLaunchKernelGGL(sw_pairs_kernel<int32_t>, dim3(n_waves), dim3(64), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col, A->vcode, A->vd_base, A->n_rows, A->view_row0, lists, status + 3);
    SW_CHECK(hipGetLastError());
    std::vector<unsigned long long> h((size_t)257 * n_waves);
    int over = 0;
    SW_CHECK(hipMemcpyAsync(&over, status + 3, sizeof over, hipMemcpyDeviceToHost, ctx->stream));
    SW_CHECK(hipMemcpyAsync(h.data(), lists, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    SW_CHECK(hipStreamSynchronize(ctx->stream));
    hipFree(lists); tmp = nullptr;
    if (over) return BIS_OK;
    std::vector<unsigned long long> &all = P.all;
    for (int w = 0; w < n_waves; ++w) {
        const size_t n = (size_t)h[(size_t)w * 257];
        if (n > (size_t)kSwMaxPairs) { all.clear(); return BIS_OK; }
        all.insert(all.end(), h.begin() + (size_t)w * 257 + 1, h.begin() + (size_t)w * 257 + 1 + n);
    }
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    P.fits = !all.empty() && all.size() <= (size_t)kSwMaxPairs;
    return BIS_OK;
}

// fmt 3 on top of a finished plan (A->sw: runs, slices, chunk offsets).  Leaves A->sw_state == 1 on success; on "does not
// apply" everything it allocated is freed and the caller goes on with another format; an out-of-memory drops the form.
static bis_status sw_try_pairs(bis_ctx *ctx, bis_mat *A, const double *table, int pad_idx, SwPairs &P) {
    bis_sellwin *sw = A->sw;
    int *status = (int *)ctx->counters + 52;
    if (!P.known) {
        if (bis_status st = sw_collect_pairs(ctx, A, P)) return st;
        if (!A->sw) return BIS_OK; // (dropped on an allocation failure)
    }
    if (!P.fits) return BIS_OK;
    const std::vector<unsigned long long> &all = P.all;
    const int n_pairs = (int)all.size();
    unsigned long long keys[256];
    double vals[256];
    int diag_pair = -1;
    for (int e = 0; e < 256; ++e) { keys[e] = ~0ull; vals[e] = 0.0; }
    for (int e = 0; e < n_pairs; ++e) {
        keys[e] = all[e];
        const unsigned vc = (unsigned)(all[e] & 0xffu);
        if (A->vd_diag && vc == 255u) {
            if (diag_pair >= 0) return BIS_OK; // diagonal entries at several offsets (a row view's numbering): another format
            diag_pair = e;
        } else vals[e] = table[vc];
    }
    vals[n_pairs] = table[pad_idx]; // the padding entry: 1.0
    const int stride = (n_pairs + 63) / 64 * 64;
    const int nb = sw->n_blocks;
    // fmt 4 (a mask per row) where the pairs fit 32 bits and the rows run through them in order, else fmt 3 (a byte per non-zero)
    // (not with per-row diagonal values: a multiplier per row and pair in vector registers measured slower than the byte codes, Anderson-256 0.140 against 0.109 ms)
    for (int attempt = (n_pairs <= 32 && diag_pair < 0 && bis_opts().spmv_sellwin_masks != 0) ? 0 : 1; attempt < 2; ++attempt) {
        const bool masks = attempt == 0;
        const int64_t total = sw->total_chunks;
        const size_t code_words = masks ? (size_t)nb * 256 * (size_t)sw->R : 64 * (size_t)(total + 1);
        int16_t *blk_base = nullptr;
        unsigned long long *pair_key = nullptr;
        uint32_t *codes = nullptr;
        double *dict3 = nullptr;
        hipError_t e = hipMalloc(&blk_base, sizeof(int16_t) * (size_t)stride * (size_t)nb);
        if (e == hipSuccess) e = hipMalloc(&pair_key, sizeof keys);
        if (e == hipSuccess) e = hipMalloc(&codes, sizeof(uint32_t) * code_words);
        if (e == hipSuccess) e = hipMalloc(&dict3, sizeof vals);
        if (e == hipSuccess) e = hipMemcpyAsync(pair_key, keys, sizeof keys, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dict3, vals, sizeof vals, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(status + 2, 0, sizeof(int), ctx->stream);
        if (e == hipSuccess && !masks) e = hipMemsetAsync(codes + (size_t)total * 64, 0, sizeof(uint32_t) * 64, ctx->stream);
        if (e == hipSuccess) {
#define SW_FILLP(RP) hipLaunchKernelGGL((sw_fill_pairs_kernel<RP>), dim3(nb), dim3(256), 0, ctx->stream, (const RP *)A->row_ptr, A->col, A->vcode, A->vd_base, A->n_rows, A->view_row0, sw->R, sw->hdr, sw->slice_chunk0, codes, pair_key, n_pairs, stride, blk_base, status)
#define SW_FILLM(RP) hipLaunchKernelGGL((sw_fill_masks_kernel<RP>), dim3(nb), dim3(256), 0, ctx->stream, (const RP *)A->row_ptr, A->col, A->vcode, A->vd_base, A->n_rows, A->view_row0, sw->R, sw->hdr, codes, pair_key, n_pairs, stride, blk_base, status)
            if (masks) { if (A->rp64) SW_FILLM(int64_t); else SW_FILLM(int32_t); }
            else { if (A->rp64) SW_FILLP(int64_t); else SW_FILLP(int32_t); }
#undef SW_FILLP
#undef SW_FILLM
            e = hipGetLastError();
        }
        int bad = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, status + 2, sizeof bad, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream); // keys[], vals[], bad
        if (e != hipSuccess || bad) {
            hipFree(blk_base); hipFree(pair_key); hipFree(codes); hipFree(dict3);
            if (e == hipSuccess && masks) continue; // the byte codes then
            if (e == hipSuccess || e == hipErrorOutOfMemory) { (void)hipGetLastError(); return BIS_OK; } // another format
            ctx->err = std::string("bis_spmv sellwin pairs: ") + hipGetErrorString(e);
            bis_spmv_sellwin_drop(A);
            A->sw_state = -1;
            return BIS_ERR_HIP;
        }
        hipFree(sw->dict);
        sw->dict = dict3;
        sw->codes = codes;
        sw->blk_base = blk_base;
        sw->pair_key = pair_key;
        sw->fmt = masks ? 4 : 3;
        sw->n_pairs = n_pairs;
        sw->pair_stride = stride;
        sw->diag_pair = diag_pair;
        A->sw_state = 1;
        return BIS_OK;
    }
    return BIS_OK;
}

// Build the form for a matrix that has a value dictionary (A->vd_state == 1); A->sw_state tells the outcome.
static bis_status sw_try_rows(bis_ctx *ctx, bis_mat *A, int R, SwPairs &P);

bis_status bis_spmv_sellwin_try(bis_ctx *ctx, bis_mat *A) {
    if (A->sw_state != 0) return BIS_OK;
    A->sw_state = -1;
    if (!sw_enabled() || A->vd_state != 1 || A->n_rows == 0 || A->nnz == 0 || A->n_cols >= ((int64_t)1 << 31) - 16) return BIS_OK;
    int R = kSwDefaultR;
    while (R > 1 && A->n_rows < (int64_t)kSwRows * R * 1024) R >>= 1; // small matrices: more, smaller blocks
    bool big = R == 2 && A->n_rows >= (int64_t)kSwRows * 4 * 1024;
    if (bis_opts().spmv_sellwin_rows > 0) {
        R = bis_opts().spmv_sellwin_rows >= 2 ? 2 : 1;
        big = bis_opts().spmv_sellwin_rows >= 4;
    }
    // Blocks of 1024 rows exist for the row-mask form only (its waves keep 4 x 4 LDS reads in flight: HPCG-256 0.116 against 0.125 ms
    // with 512 rows): tried first where that form can apply, and the plan is redone with 512 rows where it then does not.
    // The pairs are counted BEFORE that plan is made (one light pass over the matrix): a matrix with more than 32 of them goes to
    // the 512-row plan directly instead of building the 1024-row plan and its tables only to drop them.
    SwPairs P;
    if (big && !A->vd_diag && bis_opts().spmv_sellwin_masks != 0 && bis_opts().spmv_sellwin_pairs != 0) {
        if (bis_status st = sw_collect_pairs(ctx, A, P)) return st;
        if (P.fits && P.all.size() <= 32) {
            if (bis_status st = sw_try_rows(ctx, A, 4, P)) return st;
            if (A->sw_state == 1) return BIS_OK;
            A->sw_state = -1;
        }
    }
    return sw_try_rows(ctx, A, R, P);
}

static bis_status sw_try_rows(bis_ctx *ctx, bis_mat *A, int R, SwPairs &P) {
    const int64_t nb64 = (A->n_rows + (int64_t)kSwRows * R - 1) / ((int64_t)kSwRows * R);
    if (nb64 > (int64_t)1 << 26) return BIS_OK;
    const int nb = (int)nb64;
    // the table: the matrix' dictionary and the padding value 1.0
    double table[256];
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(table, A->vdict, sizeof table, hipMemcpyDeviceToHost, ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const int cap_small = A->vd_diag ? 31 : 32, cap_gen = A->vd_diag ? 255 : 256;
    int pad_idx = -1;
    for (int i = 0; i < A->vd_n; ++i)
        if (table[i] == 1.0 && !std::signbit(table[i])) pad_idx = i;
    int n_tab = A->vd_n;
    if (pad_idx < 0) {
        if (n_tab >= cap_gen) return BIS_OK; // no free code for the padding
        pad_idx = n_tab++;
        table[pad_idx] = 1.0;
    }
    bis_sellwin *sw = new bis_sellwin;
    A->sw = sw;
    sw->n_blocks = nb;
    sw->R = R;
    sw->n_slices = (int64_t)nb * 4 * R;
    sw->diag = A->vd_diag;
    // 16-bit joint codes where the table (+ the diagonal marker 7) fits 3 bits; else value bytes
    const bool joint = n_tab <= (A->vd_diag ? 7 : 8) && bis_opts().spmv_sellwin_joint != 0;
    sw->fmt = joint ? 2 : (n_tab <= cap_small ? 1 : 0);
    sw->pad_idx = pad_idx;
    sw->diag_idx = !A->vd_diag ? -1 : (sw->fmt == 2 ? 7 : (sw->fmt == 1 ? 31 : 255));
    int32_t *slice_chunks = nullptr;
    void *tmp = nullptr;
    int *status = (int *)ctx->counters + 52;
    SW_CHECK(hipMalloc(&sw->hdr, sizeof(int32_t) * 64 * (size_t)nb));
    SW_CHECK(hipMalloc(&sw->own_rank, sizeof(int32_t) * (size_t)nb));
    SW_CHECK(hipMemsetAsync(sw->own_rank, 0xFF, sizeof(int32_t) * (size_t)nb, ctx->stream)); // (blocks the plan leaves early: -1)
    SW_CHECK(hipMalloc(&slice_chunks, sizeof(int32_t) * (size_t)(sw->n_slices + 1)));
    SW_CHECK(hipMalloc(&sw->slice_chunk0, sizeof(int64_t) * (size_t)(sw->n_slices + 1)));
    SW_CHECK(hipMalloc(&sw->dict, sizeof table));
    SW_CHECK(hipMemcpyAsync(sw->dict, table, sizeof table, hipMemcpyHostToDevice, ctx->stream));
    SW_CHECK(hipMemsetAsync(status, 0, 2 * sizeof(int), ctx->stream));
    SW_CHECK(hipMemsetAsync(status + 5, 0, sizeof(int), ctx->stream)); // most chunks of a slice
    SW_CHECK(hipMemsetAsync(slice_chunks + sw->n_slices, 0, sizeof(int32_t), ctx->stream));
    if (A->rp64) hipLaunchKernelGGL(sw_plan_kernel<int64_t>, dim3(nb), dim3(256), 0, ctx->stream, (const int64_t *)A->row_ptr, A->col, A->n_rows, R, sw_gran_cap(R), A->view_row0, sw->hdr, slice_chunks, sw->own_rank, status);
    else hipLaunchKernelGGL(sw_plan_kernel<int32_t>, dim3(nb), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col, A->n_rows, R, sw_gran_cap(R), A->view_row0, sw->hdr, slice_chunks, sw->own_rank, status);
    SW_CHECK(hipGetLastError());
    int h[2] = {0, 0}, h_maxch = 0;
    SW_CHECK(hipMemcpyAsync(h, status, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    SW_CHECK(hipMemcpyAsync(&h_maxch, status + 5, sizeof h_maxch, hipMemcpyDeviceToHost, ctx->stream));
    // chunk offsets of the slices
    const int64_t ns1 = sw->n_slices + 1;
    hipLaunchKernelGGL(sw_widen_kernel, dim3((unsigned)((ns1 + 255) / 256)), dim3(256), 0, ctx->stream, slice_chunks, sw->slice_chunk0, ns1);
    SW_CHECK(hipGetLastError());
    size_t tmp_bytes = 0;
    SW_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, sw->slice_chunk0, sw->slice_chunk0, (int64_t)0, (size_t)ns1, rocprim::plus<int64_t>(), ctx->stream));
    SW_CHECK(hipMalloc(&tmp, tmp_bytes));
    SW_CHECK(rocprim::exclusive_scan(tmp, tmp_bytes, sw->slice_chunk0, sw->slice_chunk0, (int64_t)0, (size_t)ns1, rocprim::plus<int64_t>(), ctx->stream));
    int64_t total = 0;
    SW_CHECK(hipMemcpyAsync(&total, sw->slice_chunk0 + sw->n_slices, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
    SW_CHECK(hipStreamSynchronize(ctx->stream)); // table[], h, total
    hipFree(tmp); tmp = nullptr;
    hipFree(slice_chunks); slice_chunks = nullptr;
    // Every slice the same number of chunks where padding the short ones up costs < 3 % (a stencil: only the slices on the grid's
    // faces are shorter): the SpMV then walks a compile-time number of chunks
    if (h_maxch > 0 && h_maxch <= 8 && !h[0] && (double)h_maxch * (double)sw->n_slices <= 1.03 * (double)total) {
        hipLaunchKernelGGL(sw_uniform_kernel, dim3((unsigned)((ns1 + 255) / 256)), dim3(256), 0, ctx->stream, sw->slice_chunk0, ns1, (int64_t)h_maxch);
        SW_CHECK(hipGetLastError());
        total = (int64_t)h_maxch * sw->n_slices;
        sw->uniform_chunks = h_maxch;
    }
    sw->total_chunks = total;
    sw->max_gran = h[1];
    // not representable, or more than 30 % of padding (ragged rows): the gather kernels stay
    if (h[0] || (double)total * 256.0 > 1.3 * (double)A->nnz + 256.0 * 4 * 64) {
        bis_spmv_sellwin_drop(A);
        A->sw_state = -1;
        return BIS_OK;
    }
    // fmt 3 first: one byte per non-zero where the matrix has few (column - row, value) pairs that run through every
    // block's window in step with the rows
    if (bis_opts().spmv_sellwin_pairs != 0) {
        if (bis_status st = sw_try_pairs(ctx, A, table, pad_idx, P)) return st;
        if (A->sw && sw->R == 4 && !(A->sw_state == 1 && sw->fmt == 4)) { // (blocks of 1024 rows exist for the row-mask form only)
            bis_spmv_sellwin_drop(A);
            A->sw_state = -1;
            return BIS_OK;
        }
        if (A->sw_state == 1 || !A->sw) return BIS_OK;
    }
    if (sw->R == 4) { bis_spmv_sellwin_drop(A); A->sw_state = -1; return BIS_OK; }
    const size_t cw = sw->fmt == 2 ? 128 : 192; // 32-bit words per chunk of 64 lanes
    SW_CHECK(hipMalloc(&sw->codes, sizeof(uint32_t) * cw * (size_t)(total + 1)));
    SW_CHECK(hipMemsetAsync(sw->codes + (size_t)total * cw, 0, sizeof(uint32_t) * cw, ctx->stream));
#define SW_FILL(RP) hipLaunchKernelGGL((sw_fill_kernel<RP>), dim3(nb), dim3(256), 0, ctx->stream, (const RP *)A->row_ptr, A->col, A->vcode, A->vd_base, A->n_rows, R, sw->fmt, sw->hdr, sw->slice_chunk0, sw->codes, sw->pad_idx, sw->diag_idx)
    if (A->rp64) SW_FILL(int64_t); else SW_FILL(int32_t);
#undef SW_FILL
    SW_CHECK(hipGetLastError());
    A->sw_state = 1;
    return BIS_OK;
}

int bis_spmv_sellwin_blocks(const bis_mat *A) { return A->sw_state == 1 ? A->sw->n_blocks : 0; }
int64_t bis_spmv_sellwin_slices(const bis_mat *A) { return A->sw_state == 1 ? A->sw->n_slices : 0; }
int bis_spmv_sellwin_format(const bis_mat *A) { return A->sw_state == 1 ? A->sw->fmt : -1; }

// bytes of the form's own arrays one launch reads: the code stream (with its padding), block headers, slice offsets, table
int64_t bis_spmv_sellwin_bytes(const bis_mat *A) {
    if (A->sw_state != 1) return 0;
    if (A->sw->fmt == 4) // a mask per row, the block headers and bases, the pairs' values
        return 4 * (int64_t)A->sw->n_blocks * 256 * A->sw->R + (int64_t)A->sw->n_blocks * (256 + 2 * A->sw->pair_stride + 4) + 2048;
    const int64_t chunk_bytes = A->sw->fmt == 3 ? 256 : (A->sw->fmt == 2 ? 512 : 768);
    return A->sw->total_chunks * chunk_bytes + (int64_t)A->sw->n_blocks * (256 + 2 * A->sw->pair_stride) + 8 * (A->sw->n_slices + 1) + 2048;
}

// mode 0 / 1 as in the kernel; grid and remap_arg from the caller's block map over bis_spmv_sellwin_blocks(A)
bis_status bis_spmv_sellwin_launch(bis_ctx *ctx, const bis_mat *A, const double *x, double *y, int mode, const double *w,
                                   double *partials, const int *stop, int remap_arg, int grid) {
    const bis_sellwin *sw = A->sw;
    const int x_al16 = ((uintptr_t)x & 15) == 0;
    const size_t win = (size_t)(2 + 8 * sw->max_gran) * 8;
    const size_t win_off3 = (size_t)sw_layout3(sw->n_pairs, sw->diag, sw->R).win_off;
    // diagnostic (BIS_SELLWIN_DEBUG=file): per-block cycle stamps of the last launch -- start, header arrived, barrier passed, end
    static long long *dbg_buf = nullptr;
    static int dbg_cap = 0;
    long long *dbg = nullptr;
    const char *dbg_file = getenv("BIS_SELLWIN_DEBUG");
    if (dbg_file) {
        if (dbg_cap < sw->n_blocks) { hipFree(dbg_buf); dbg_buf = nullptr; if (hipMalloc(&dbg_buf, sizeof(long long) * 4 * (size_t)sw->n_blocks) == hipSuccess) dbg_cap = sw->n_blocks; }
        dbg = dbg_buf;
    }
    const int32_t *own = (mode == 1 && w == x + A->view_row0 && x_al16) ? sw->own_rank : nullptr; // the fused dot's w is x itself (CG: p)
    if (sw->fmt == 4) {
#define SM_L3(MODE, RR, NN)                                                                                                \
    hipLaunchKernelGGL((spmv_sellmask_kernel<MODE, RR, NN>), dim3(grid), dim3(256), win, ctx->stream, x, y, A->n_rows, A->n_cols, \
                       sw->n_blocks, remap_arg, w, partials, stop, sw->hdr, sw->codes, sw->dict, x_al16 | (bis_opts().spmv_sellwin_nt != 0 ? 2 : 0), \
                       (const int32_t *)sw->blk_base, sw->n_pairs, own, dbg)
#define SM_L2(MODE) do {                                                                                                   \
        if (sw->R == 4 && sw->n_pairs == 27) SM_L3(MODE, 4, 27);                                                           \
        else if (sw->R == 4) SM_L3(MODE, 4, 0);                                                                            \
        else if (sw->R == 2 && sw->n_pairs == 27) SM_L3(MODE, 2, 27);                                                      \
        else if (sw->R == 2) SM_L3(MODE, 2, 0); else SM_L3(MODE, 1, 0); } while (0)
        if (mode == 1) SM_L2(1); else SM_L2(0);
#undef SM_L2
#undef SM_L3
    } else {
#define SW_L4(MODE, DIAG, FMT, RR)                                                                                     \
    hipLaunchKernelGGL((spmv_sellwin_kernel<MODE, DIAG, FMT, RR>), dim3(grid), dim3(256), (FMT == 3 ? win_off3 : (size_t)(SwLayout<DIAG, FMT, RR>::kWinOff)) + win, \
                       ctx->stream, x, y, A->n_rows, A->n_cols, sw->n_blocks, remap_arg, w, partials, stop, sw->hdr,   \
                       sw->slice_chunk0, sw->codes, sw->dict, A->vdiag, x_al16 | (bis_opts().spmv_sellwin_nt != 0 ? 2 : 0), sw->blk_base, sw->pair_stride, sw->n_pairs, sw->diag_pair, own, dbg)
#define SW_L4N(MODE, DIAG, RR, NN)                                                                                      \
    hipLaunchKernelGGL((spmv_sellwin_kernel<MODE, DIAG, 3, RR, NN>), dim3(grid), dim3(256), win_off3 + win,             \
                       ctx->stream, x, y, A->n_rows, A->n_cols, sw->n_blocks, remap_arg, w, partials, stop, sw->hdr,   \
                       sw->slice_chunk0, sw->codes, sw->dict, A->vdiag, x_al16 | (bis_opts().spmv_sellwin_nt != 0 ? 2 : 0), sw->blk_base, sw->pair_stride, sw->n_pairs, sw->diag_pair, own, dbg)
#define SW_L3(MODE, DIAG, FMT) do {                                                                                    \
        if (FMT == 3 && sw->uniform_chunks == 7 && sw->R == 2) SW_L4N(MODE, DIAG, 2, 7);                               \
        else if (FMT == 3 && sw->uniform_chunks == 2 && sw->R == 2) SW_L4N(MODE, DIAG, 2, 2);                          \
        else if (sw->R == 2) SW_L4(MODE, DIAG, FMT, 2); else SW_L4(MODE, DIAG, FMT, 1); } while (0)
#define SW_L2(MODE, DIAG) do { if (sw->fmt == 3) SW_L3(MODE, DIAG, 3); else if (sw->fmt == 2) SW_L3(MODE, DIAG, 2); else if (sw->fmt == 1) SW_L3(MODE, DIAG, 1); else SW_L3(MODE, DIAG, 0); } while (0)
#define SW_L1(MODE) do { if (sw->diag) SW_L2(MODE, true); else SW_L2(MODE, false); } while (0)
    if (mode == 1) SW_L1(1); else SW_L1(0);
#undef SW_L1
#undef SW_L2
#undef SW_L3
#undef SW_L4
#undef SW_L4N
    }
    BIS_HIP_CHECK(ctx, hipGetLastError());
    if (dbg) {
        std::vector<long long> hd((size_t)4 * sw->n_blocks);
        BIS_HIP_CHECK(ctx, hipMemcpyAsync(hd.data(), dbg, sizeof(long long) * hd.size(), hipMemcpyDeviceToHost, ctx->stream));
        BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (FILE *f = fopen(dbg_file, "w")) {
            for (int b = 0; b < sw->n_blocks; ++b) fprintf(f, "%lld %lld %lld %lld\n", hd[4 * b], hd[4 * b + 1], hd[4 * b + 2], hd[4 * b + 3]);
            fclose(f);
        }
    }
    return BIS_OK;
}

// =====================================================================================================================
// win8: the SAME window + sliced-ELL plan for matrices with ARBITRARY values (round 5).  Per non-zero the kernel streams the
// 8-byte CRS value and a 2-byte window slot -- 10 bytes, like the packed row-block kernel of bis_spmv.hip -- but x comes
// from the block's LDS window (every entry once per block, by LDS-DMA) instead of one 8-byte gather per non-zero through
// the texture path, a lane owns a row (no product staging in LDS, no row_ptr, no second pass), and the stream is laid out
// in the order the lanes consume it: a chunk = 4 entries of each of the 64 rows of a slice = 2560 contiguous bytes
// [64 x (4 x 16-bit slot)][64 x 2 values][64 x 2 values], read with one 8-byte and two 16-byte non-temporal loads per lane.
// Entry j of a row is entry j of the CRS row; product and sum are rounded separately, in CRS order: y is bit-identical to
// the other kernels'.  Padding: value 1.0 at the slot that holds -0.0.  The CRS arrays stay authoritative (the stream is a
// lossless re-layout of val, built on the device at the first SpMV and dropped when values change).
// Usable by any matrix whose blocks of 256 R rows read at most 64 runs / 60 KB of x (runs a few granules apart are merged: banded /
// stencil orderings, RCM-ordered meshes); the rows of a block are taken in order of their length (w8_plan_kernel), so rows of
// very different lengths do not pad each other (refused beyond 12 % padding).
namespace {

constexpr int kW8ChunkBytes = 2560;


constexpr int kW8Runs = 64;     // runs of the window per block at most (header: 2 x 64 words)
constexpr int kW8GapMerge = 4;  // runs at most this many granules apart become one (the gap's x entries are copied too)

// win8's own plan, one workgroup per block of 256 R rows:
//  * the 8-column granules of x the block reads (LDS hash set, bitonic sort), cut into runs of consecutive granules; runs at most
//    kW8GapMerge granules apart are MERGED (an RCM-ordered mesh reads many short runs with small gaps: 34 raw runs per 1024-row
//    block become 23), runs of more than 64 granules split (a wave copies 64 x 64 bytes per instruction); at most kW8Runs runs
//    and max_gran granules (gaps included), else status[0] = 1;
//  * the block's rows SORTED BY LENGTH, longest first (stable: equal lengths keep their order), 64 sorted positions = one slice:
//    a slice is padded to ITS longest row, so rows of very different lengths no longer pad each other (FEM-like rows of 18-81
//    entries: 19 % padding -> 4 %).  row_of[b * 256 R + position] = row in the block; own_rank[b] >= 0 only where the order is
//    the identity (every row as long as the first, or already descending) and the own columns lie in the window side by side.
template <typename RP>
__global__ __launch_bounds__(256) void w8_plan_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col, int64_t n_rows, int R,
                                                      int max_gran, int64_t row0, int32_t *__restrict__ hdr, int32_t *__restrict__ slice_chunks,
                                                      int32_t *__restrict__ own_rank, uint16_t *__restrict__ row_of, int *status) {
    __shared__ int table[kSwHash];
    __shared__ int list[1024];
    __shared__ int keys[1024]; // (length << 10 | 1023 - row) of the block's rows: descending order = longest first, equal lengths by row
    __shared__ int cnt, failed, n_out, win_gran, own_s, ident;
    __shared__ int out_g0[kW8Runs], out_rk[kW8Runs];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t r0 = (int64_t)b * kSwRows * R;
    const int rows = (int)min((int64_t)kSwRows * R, n_rows - r0);
    const int cap = kSwRows * R; // 256, 512 or 1024
    for (int i = tid; i < kSwHash; i += 256) table[i] = -1;
    if (tid == 0) { cnt = 0; failed = 0; n_out = 0; win_gran = 0; own_s = -1; ident = 1; }
    __syncthreads();
    // rows by length
    for (int i = tid; i < 1024; i += 256) {
        int len = -1; // (positions past the block's rows sort behind every row)
        if (i < rows) len = (int)((int64_t)row_ptr[r0 + i + 1] - (int64_t)row_ptr[r0 + i]);
        keys[i] = i < cap ? ((min(len, (1 << 20) - 1) + 1) << 10 | (1023 - i)) : -1;
    }
    __syncthreads();
    for (int k = 2; k <= 1024; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < 1024; i += 256) {
                const int o = i ^ j;
                if (o > i) {
                    const int a = keys[i], c = keys[o];
                    if ((a < c) == ((i & k) == 0)) { keys[i] = c; keys[o] = a; } // descending
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < cap; i += 256) {
        const int row = 1023 - (keys[i] & 1023);
        row_of[(size_t)b * cap + i] = (uint16_t)row;
        if (row != i) atomicExch(&ident, 0);
        if ((i & 63) == 0) { // first (= longest) row of slice i / 64
            const int len = (keys[i] >> 10) - 1;
            const int ch = len > 0 ? (len + 3) >> 2 : 0;
            slice_chunks[(size_t)b * 4 * R + (i >> 6)] = ch;
            atomicMax(&status[5], ch);
        }
    }
    // granules
    const int64_t s = (int64_t)row_ptr[r0], e = (int64_t)row_ptr[r0 + rows];
    for (int64_t k = s + tid; k < e; k += 256) {
        if (__hip_atomic_load(&failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        const int g = col[k] >> 3;
        unsigned h = ((unsigned)g * 2654435761u) >> 20;
        for (;;) {
            const int cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (cur == g) break;
            if (cur == -1) {
                const int old = atomicCAS(&table[h], -1, g);
                if (old == -1) {
                    if (atomicAdd(&cnt, 1) >= max_gran) atomicExch(&failed, 1);
                    break;
                }
                if (old == g) break;
            }
            h = (h + 1) & (kSwHash - 1);
        }
    }
    __syncthreads();
    const int n = cnt;
    if (failed || n > max_gran) {
        if (tid == 0) atomicExch(&status[0], 1);
        return;
    }
    __syncthreads();
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (int i = tid; i < kSwHash; i += 256) {
        const int g = table[i];
        if (g != -1) list[atomicAdd(&cnt, 1)] = g;
    }
    int P = 2;
    while (P < n) P <<= 1;
    __syncthreads();
    for (int i = n + tid; i < P; i += 256) list[i] = INT32_MAX;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += 256) {
                const int o = i ^ j;
                if (o > i) {
                    const int a = list[i], c = list[o];
                    if ((a > c) == ((i & k) == 0)) { list[i] = c; list[o] = a; }
                }
            }
            __syncthreads();
        }
    if (tid == 0) { // one thread walks the sorted granules: merge, split, rank (a setup kernel: ~n steps)
        int m = 0, rank = 0; // runs emitted, granules of the window so far (gaps included)
        int i = 0;
        bool ok = true;
        const int64_t c_first = r0 + row0, c_last = c_first + rows - 1;
        const int g_first = (int)(c_first >> 3), g_last = (int)(c_last >> 3);
        int own = -1;
        while (i < n && ok) {
            const int g0 = list[i];
            int g1 = g0; // last granule of the merged run
            ++i;
            while (i < n && list[i] - g1 <= kW8GapMerge + 1) { g1 = list[i]; ++i; }
            const int len = g1 - g0 + 1;
            if ((c_first & 7) == 0 && g0 <= g_first && g_last <= g1) own = rank + (g_first - g0);
            for (int o = 0; o < len && ok; o += kSwRunGran) {
                if (m >= kW8Runs) { ok = false; break; }
                out_g0[m] = g0 + o;
                out_rk[m] = (rank + o) | (min(kSwRunGran, len - o) << 16);
                ++m;
            }
            rank += len;
            if (rank > max_gran) ok = false;
        }
        if (!ok) atomicExch(&failed, 1);
        n_out = m;
        win_gran = rank;
        own_s = own;
    }
    __syncthreads();
    if (failed) {
        if (tid == 0) atomicExch(&status[0], 1);
        return;
    }
    if (tid < 2 * kW8Runs) {
        const int k = tid % kW8Runs;
        int word = 0;
        if (k < n_out) word = tid < kW8Runs ? out_g0[k] : out_rk[k];
        hdr[(size_t)b * 2 * kW8Runs + tid] = word;
    }
    if (tid == 0) {
        own_rank[b] = ident ? own_s : -1;
        atomicMax(&status[1], win_gran);
    }
}

template <typename RP>
__global__ __launch_bounds__(256) void w8_fill_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col, const double *__restrict__ val,
                                                      int64_t n_rows, int R, const int32_t *__restrict__ hdr,
                                                      const int64_t *__restrict__ slice_chunk0, const uint16_t *__restrict__ row_of,
                                                      unsigned char *__restrict__ stream) {
    __shared__ int g0s[kW8Runs], rk[kW8Runs];
    __shared__ int nr_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < kW8Runs) {
        g0s[tid] = hdr[(size_t)b * 2 * kW8Runs + tid];
        const int w2 = hdr[(size_t)b * 2 * kW8Runs + kW8Runs + tid];
        rk[tid] = w2 & 0xffff;
        const unsigned long long m = __ballot((w2 >> 16) != 0);
        if (tid == 0) nr_s = __popcll(m);
    }
    __syncthreads();
    const int nr = nr_s;
    for (int rr = 0; rr < R; ++rr) {
        const int64_t slice = ((int64_t)b * 4 + wv) * R + rr;
        const int64_t pos = slice * 64 + lane; // position in the block's length order
        const int64_t r = (int64_t)b * kSwRows * R + row_of[pos];
        int64_t rs = 0;
        int len = 0;
        if (r < n_rows) {
            rs = (int64_t)row_ptr[r];
            len = (int)((int64_t)row_ptr[r + 1] - rs);
        }
        const int64_t c0 = slice_chunk0[slice];
        const int nch = (int)(slice_chunk0[slice + 1] - c0);
        for (int c = 0; c < nch; ++c) {
            unsigned cc[4];
            double vv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 4 * c + q;
                cc[q] = 0;      // the -0.0 slot
                vv[q] = 1.0;    // 1.0 * -0.0 = -0.0: acc + -0.0 == acc for every acc
                if (j < len) {
                    const int ci = col[rs + j];
                    const int g = ci >> 3;
                    int lo = 0, hi = nr - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (g0s[mid] <= g) lo = mid; else hi = mid - 1;
                    }
                    cc[q] = (unsigned)(2 + (rk[lo] + (g - g0s[lo])) * 8 + (ci & 7)) * 8u; // byte offset of the slot behind the window's base
                    vv[q] = val[rs + j];
                }
            }
            unsigned char *p = stream + (size_t)(c0 + c) * kW8ChunkBytes;
            *reinterpret_cast<uint2 *>(p + lane * 8) = make_uint2(cc[0] | cc[1] << 16, cc[2] | cc[3] << 16);
            *reinterpret_cast<double2 *>(p + 512 + lane * 16) = make_double2(vv[0], vv[1]);
            *reinterpret_cast<double2 *>(p + 1536 + lane * 16) = make_double2(vv[2], vv[3]);
        }
    }
}

typedef double w8_v2d __attribute__((ext_vector_type(2)));
typedef unsigned w8_v2u __attribute__((ext_vector_type(2)));
struct W8Chunk { w8_v2u code; w8_v2d v0, v1; };

// one chunk of the stream for this lane: three non-temporal loads (the stream is read once per product)
__device__ __forceinline__ W8Chunk w8_load(const unsigned char *__restrict__ stream, int64_t c, int lane) {
    const unsigned char *p = stream + (size_t)c * kW8ChunkBytes;
    W8Chunk ch;
    ch.code = __builtin_nontemporal_load(reinterpret_cast<const w8_v2u *>(p + lane * 8));
    ch.v0 = __builtin_nontemporal_load(reinterpret_cast<const w8_v2d *>(p + 512 + lane * 16));
    ch.v1 = __builtin_nontemporal_load(reinterpret_cast<const w8_v2d *>(p + 1536 + lane * 16));
    return ch;
}

// acc += v_q * window[slot_q], q = 0..3, products and sums rounded separately (the CRS kernels' arithmetic)
__device__ __forceinline__ void w8_consume(const unsigned char *win, const W8Chunk &ch, double &acc) {
#pragma clang fp contract(off)
    const double x0 = *reinterpret_cast<const double *>(win + (ch.code.x & 0xffffu));
    const double x1 = *reinterpret_cast<const double *>(win + (ch.code.x >> 16));
    const double x2 = *reinterpret_cast<const double *>(win + (ch.code.y & 0xffffu));
    const double x3 = *reinterpret_cast<const double *>(win + (ch.code.y >> 16));
    const double p0 = ch.v0.x * x0, p1 = ch.v0.y * x1, p2 = ch.v1.x * x2, p3 = ch.v1.y * x3;
    acc = acc + p0;
    acc = acc + p1;
    acc = acc + p2;
    acc = acc + p3;
}

// MODE 0: y = A x.  MODE 1: also partials[4 b + wave] = sum over the wave's rows of y[r] w[r] (CG's (Ap, p)).
// A wave owns the R consecutive slices slice0 .. slice0 + R - 1; their chunks are ONE contiguous range of the stream (the chunk
// offsets are a prefix sum), walked with a ring of D chunks in flight per lane: the chunk D places ahead is requested as soon
// as a ring entry has been consumed, so the stream never waits for the arithmetic, and the first D chunks are requested before
// the window is (they do not need it).  LDS: 16 bytes (the -0.0 slot), then the window.
template <int MODE, int R, int D>
__global__ __launch_bounds__(256) void spmv_win8_kernel(
    const double *x, double *__restrict__ y, int64_t n_rows, int64_t n_cols, int n_blocks, int remap_arg, const double *w,
    double *__restrict__ partials, const int *stop, const int32_t *__restrict__ hdr, const int64_t *__restrict__ slice_chunk0,
    const unsigned char *__restrict__ stream, int x_al16, const int32_t *__restrict__ own_rank, const uint16_t *__restrict__ row_of) {
    if (stop && stop[1]) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int b = remap_arg > 0 ? xcd_remap(blockIdx.x, remap_arg)
                                : (remap_arg < -1 ? xcd_group_remap(blockIdx.x, -remap_arg) : (int)blockIdx.x);
    if (b >= n_blocks) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw_g = hdr[(size_t)b * 2 * kW8Runs + lane], hw_r = hdr[(size_t)b * 2 * kW8Runs + kW8Runs + lane]; // run k: first granule / rank | granules << 16
    const int64_t slice0 = ((int64_t)b * 4 + wv) * R;
    const int64_t block_row0 = (int64_t)b * kSwRows * R;
    int64_t rows_of[R]; // the rows this lane owns: position (slice, lane) of the block's length order -> row
#pragma unroll
    for (int r = 0; r < R; ++r) rows_of[r] = block_row0 + row_of[(slice0 + r) * 64 + lane];
    // chunk boundaries of the wave's slices: lane r holds slice_chunk0[slice0 + r], r <= R
    const int64_t my_bnd = slice_chunk0[slice0 + min(lane, R)];
    auto bnd = [&](int r) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)my_bnd, r);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)my_bnd >> 32), r);
        return (int64_t)(((unsigned long long)hi << 32) | lo);
    };
    const int64_t C0 = bnd(0), C1 = bnd(R);
    const int64_t c_last = max(C1 - 1, C0); // (the stream ends with one spare chunk: an empty wave reads it)
    W8Chunk ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = w8_load(stream, min(C0 + d, c_last), lane);
    // MODE 1: the dot's operand.  own_rank != nullptr: w is x at the rows' own columns (CG: w = x = p) -- where the block's window
    // holds them side by side (own >= 0) the operand comes from LDS behind the barrier instead of a second global read of p
    double wr[R];
    int own = -1;
    if (MODE == 1 && own_rank) own = own_rank[b];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = rows_of[r];
        wr[r] = (MODE == 1 && own < 0 && row < n_rows) ? w[row] : 0.0;
    }
    if (tid == 0) *reinterpret_cast<double *>(lds) = -0.0;
    {   // window: the runs of 8-column granules, 16-byte pieces by LDS-DMA (as spmv_sellwin_kernel)
        const int n_runs = __popcll(__ballot((hw_r >> 16) != 0));
        unsigned char *win = lds + 16;
        for (int k = 0; k < n_runs; ++k) {
            const int g0 = __builtin_amdgcn_readlane(hw_g, k);
            const int w2 = __builtin_amdgcn_readlane(hw_r, k);
            const int rank = w2 & 0xffff, n_pieces = (w2 >> 16) * 4;
            const int j = (wv + k) & 3;
            const int p = j * 64 + lane;
            const int64_t c = (int64_t)g0 * 8 + 2 * p;
            unsigned char *dst = win + (size_t)rank * 64 + (size_t)j * 1024;
            if (p < n_pieces) {
                if (x_al16 && c + 1 < n_cols) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + c),
                                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                } else {
                    double2 v;
                    v.x = c < n_cols ? x[c] : 0.0;
                    v.y = c + 1 < n_cols ? x[c + 1] : 0.0;
                    *reinterpret_cast<double2 *>(dst + lane * 16) = v;
                }
            }
        }
    }
    __syncthreads();
    if (MODE == 1 && own >= 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) wr[r] = *reinterpret_cast<const double *>(lds + 16 + ((size_t)own * 8 + (wv * R + r) * 64 + lane) * 8);
    }
    double dot_acc = 0.0; // MODE 1: sum over the wave's slices of y[row] w[row] (one partial per WAVE)
    // the walk over [C0, C1): ring[d] holds chunk c whenever (c - C0) % D == d, so the ring index is a compile-time constant in
    // the unrolled round; slice rr of the wave ends at chunk e[rr] (an empty slice: e[rr] == e[rr - 1]).  All slice indices
    // are compile-time constants too (a run-time index put wr[] into scratch memory).
    int64_t e[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) e[rr] = bnd(rr + 1);
    int64_t c = C0;
    int next = 0; // the slice being summed (wave-uniform)
    double acc = 0.0;
#define W8_END_SLICES()                                                                    \
    _Pragma("unroll") for (int rr = 0; rr < R; ++rr)                                       \
        if (rr == next && c == e[rr]) {                                                    \
            const int64_t row = rows_of[rr];                                               \
            if (row < n_rows) y[row] = acc;                                                \
            if (MODE == 1) dot_acc += row < n_rows ? acc * wr[rr] : 0.0;                   \
            acc = 0.0;                                                                     \
            ++next;                                                                        \
        }
    W8_END_SLICES()
    while (c < C1) { // (wave-uniform)
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (c < C1) {
                const W8Chunk cur = ring[d];
                ring[d] = w8_load(stream, min(c + D, c_last), lane);
                w8_consume(lds, cur, acc);
                ++c;
                W8_END_SLICES()
            }
        }
    }
#undef W8_END_SLICES
    if (MODE == 1) {
        const double t = wave_sum(dot_acc);
        if (lane == 0) partials[(size_t)b * 4 + wv] = t;
    }
}

} // namespace

void bis_spmv_win8_drop(bis_mat *A) {
    if (A->sw8) {
        if (A->sw8->own_codes) A->sw8->codes = reinterpret_cast<uint32_t *>(A->sw8->own_codes); // (a redirected stream is the caller's memory: never freed here)
        hipFree(A->sw8->hdr); hipFree(A->sw8->slice_chunk0); hipFree(A->sw8->own_rank); hipFree(A->sw8->codes); hipFree(A->sw8->row_of);
        delete A->sw8;
        A->sw8 = nullptr;
    }
    A->sw8_state = 0;
}

int bis_spmv_win8_blocks(const bis_mat *A) { return A->sw8_state == 1 ? A->sw8->n_blocks : 0; }
int64_t bis_spmv_win8_slices(const bis_mat *A) { return A->sw8_state == 1 ? A->sw8->n_slices : 0; }
int64_t bis_spmv_win8_partials(const bis_mat *A) { return A->sw8_state == 1 ? (int64_t)A->sw8->n_blocks * 4 : 0; } // fused dot: one per wave
// bytes of the form's own arrays one launch reads: the stream (with its padding), block headers, slice offsets
int64_t bis_spmv_win8_bytes(const bis_mat *A) {
    if (A->sw8_state != 1) return 0;
    return A->sw8->total_chunks * (int64_t)kW8ChunkBytes + (int64_t)A->sw8->n_blocks * (8 * kW8Runs + 2 * kSwRows * A->sw8->R) + 8 * (A->sw8->n_slices + 1);
}

// placement tuning (bis_mat_tune_placement): the stream's size, and an exchange of the buffer the kernel reads
size_t bis_spmv_win8_stream_bytes(const bis_mat *A) { return A->sw8_state == 1 ? (size_t)kW8ChunkBytes * (size_t)(A->sw8->total_chunks + 1) : 0; }
void *bis_spmv_win8_swap_stream(bis_mat *A, void *stream) {
    void *old = A->sw8->codes;
    A->sw8->codes = reinterpret_cast<uint32_t *>(stream);
    return old;
}

#define W8_CHECK(call)                                                         \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) {                                                \
            (void)hipGetLastError();                                           \
            hipFree(slice_chunks); hipFree(tmp);                               \
            bis_spmv_win8_drop(A);                                             \
            A->sw8_state = -1;                                                 \
            if (e_ == hipErrorOutOfMemory) return BIS_OK; /* the gather kernel stays */ \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);      \
            return BIS_ERR_HIP;                                                \
        }                                                                      \
    } while (0)

// Build the form (A->sw8_state: 1 usable, -1 does not apply).  Rows per lane: option spmv_win8_rows (1, 2, 4), default 2 for
// matrices of at least half a million rows.
// Placement tuning of the stream at build time.  WHERE in HBM the stream lies decides between two levels of the kernel's time,
// 13 % apart (HPCG-256: 0.755 / 0.855 ms; constant over time for an allocation, independent of where x and y lie, and the slow
// level is the common one early in a process: tools/win8_place2.py, tools/win8_timeline.py, profiles/r05_f_win8_placement.log).
// So a stream of 1 GiB or more is tried in up to k fresh allocations (option spmv_win8_tune, default 12 -- five slow allocations
// in a row have been seen --; the earlier ones are
// held so that the next one lands elsewhere; bounded by the free memory minus 8 GiB), each a device-to-device copy timed with
// the kernel itself on a zero vector, and the search ends at the first allocation of the fast level.
static bis_status w8_tune_placement(bis_ctx *ctx, bis_mat *A) {
    bis_sellwin *sw = A->sw8;
    const size_t bytes = bis_spmv_win8_stream_bytes(A);
    const int k = bis_opts().spmv_win8_tune >= 0 ? bis_opts().spmv_win8_tune : (bytes >= ((size_t)1 << 30) ? 12 : 0);
    if (k <= 0) return BIS_OK;
    double *x = nullptr, *y = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<void *> losers;
    auto cleanup = [&]() {
        hipStreamSynchronize(ctx->stream);
        for (void *l : losers) hipFree(l);
        hipFree(x); hipFree(y);
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
        (void)hipGetLastError();
    };
    if (hipMalloc(&x, sizeof(double) * (size_t)(A->n_cols + 2)) != hipSuccess || hipMalloc(&y, sizeof(double) * (size_t)A->n_rows) != hipSuccess ||
        hipMemsetAsync(x, 0, sizeof(double) * (size_t)(A->n_cols + 2), ctx->stream) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { cleanup(); return BIS_OK; } // no room to tune: keep the first allocation
    const int nb = sw->n_blocks, remap_arg = bis_spmv_remap_arg(nb), grid = bis_spmv_grid(nb);
    auto measure = [&](double &ms) -> bool {
        for (int i = 0; i < 2; ++i) if (bis_spmv_win8_launch(ctx, A, x, y, 0, nullptr, nullptr, nullptr, remap_arg, grid) != BIS_OK) return false;
        hipEventRecord(e0, ctx->stream);
        for (int i = 0; i < 5; ++i) if (bis_spmv_win8_launch(ctx, A, x, y, 0, nullptr, nullptr, nullptr, remap_arg, grid) != BIS_OK) return false;
        hipEventRecord(e1, ctx->stream);
        if (hipEventSynchronize(e1) != hipSuccess) return false;
        float f = 0.f;
        hipEventElapsedTime(&f, e0, e1);
        ms = f / 5.0;
        return true;
    };
    double best = 0.0;
    if (!measure(best)) { cleanup(); ctx->err = "win8 placement tuning: launch failed"; return BIS_ERR_HIP; }
    sw->tune_first_ms = best;
    // the search ends at an allocation of the fast level: one the kernel reads at >= 5.9 TB/s (fast: 6.0-6.5, slow: 5.5 on uniform
    // rows), or one at least 8 % faster than the slowest seen (ragged rows never reach 5.9)
    double slowest = best;
    auto fast_enough = [&](double ms) { return (double)bytes / (ms * 1e-3) >= 5.9e12 || ms <= 0.92 * slowest; };
    int trials = 0;
    for (; trials < k && !fast_enough(best); ++trials) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + ((size_t)8 << 30)) break;
        void *cand = nullptr;
        if (hipMalloc(&cand, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        void *cur = bis_spmv_win8_swap_stream(A, cand);
        hipMemcpyAsync(cand, cur, bytes, hipMemcpyDeviceToDevice, ctx->stream);
        double ms = 0.0;
        if (!measure(ms)) { bis_spmv_win8_swap_stream(A, cur); losers.push_back(cand); cleanup(); ctx->err = "win8 placement tuning: launch failed"; return BIS_ERR_HIP; }
        slowest = std::max(slowest, ms);
        if (ms < best) { best = ms; losers.push_back(cur); }
        else { bis_spmv_win8_swap_stream(A, cur); losers.push_back(cand); }
    }
    sw->tune_trials = trials;
    sw->tune_kept_ms = best;
    if (getenv("BIS_WIN8_STATS")) fprintf(stderr, "win8 placement: %d re-allocation(s) of %zu bytes tried, kernel %.4f ms on the first allocation, %.4f ms on the one kept\n", trials, bytes, sw->tune_first_ms, best);
    cleanup();
    return BIS_OK;
}

// debugging / tuning aid (tools/win8_offsets.py): the stream's address and size; *set != NULL: read the stream from there from now
// on (the caller owns that memory and has copied the stream into it; the library's own buffer stays allocated)
extern "C" BIS_API bis_status bis_mat_win8_debug_stream(bis_mat *A, void **ptr, size_t *bytes, void *set) {
    if (!A || A->sw8_state != 1) return BIS_ERR_INVALID;
    bis_sellwin *sw = A->sw8;
    if (ptr) *ptr = sw->own_codes ? sw->own_codes : (void *)sw->codes;
    if (bytes) *bytes = bis_spmv_win8_stream_bytes(A);
    if (set) { if (!sw->own_codes) sw->own_codes = (void *)sw->codes; sw->codes = reinterpret_cast<uint32_t *>(set); }
    else if (sw->own_codes) { sw->codes = reinterpret_cast<uint32_t *>(sw->own_codes); sw->own_codes = nullptr; }
    return BIS_OK;
}

extern "C" BIS_API void bis_mat_win8_tuning(const bis_mat *A, int *trials, double *first_ms, double *kept_ms) {
    const bool ok = A && A->sw8_state == 1;
    if (trials) *trials = ok ? A->sw8->tune_trials : 0;
    if (first_ms) *first_ms = ok ? A->sw8->tune_first_ms : 0.0;
    if (kept_ms) *kept_ms = ok ? A->sw8->tune_kept_ms : 0.0;
}

static bis_status w8_try_rows(bis_ctx *ctx, bis_mat *A, int R, bool *window_too_large);

bis_status bis_spmv_win8_try(bis_ctx *ctx, bis_mat *A) {
    if (A->sw8_state != 0) return BIS_OK;
    A->sw8_state = -1;
    if (A->n_rows == 0 || A->nnz == 0 || A->n_cols >= ((int64_t)1 << 31) - 16) return BIS_OK;
    // default 4 rows per lane (blocks of 1024 rows): HPCG-256 0.78 ms against 0.86 with 2 and 1.20 with 1 -- the larger the block, the
    // fewer times an x entry is copied into some block's window (tools/win8_probe.py, profiles/r05_c_win8_probe.log).  Where a
    // block of that size reads more than 64 runs / 60 KB of x the plan is made again with half the rows (an RCM-ordered mesh of
    // 1.5 M rows: not representable with 1024 rows, 43.6 KB windows with 512: 0.193 against 0.237 ms for the row-block kernel).
    int R = bis_opts().spmv_win8_rows > 0 ? std::min(bis_opts().spmv_win8_rows, 4) : 4;
    if (R == 3) R = 2;
    while (R > 1 && A->n_rows < (int64_t)kSwRows * R * 1024) R >>= 1;
    for (;; R >>= 1) {
        bool too_large = false;
        if (bis_status st = w8_try_rows(ctx, A, R, &too_large)) return st;
        if (A->sw8_state == 1 || !too_large || R == 1 || bis_opts().spmv_win8_rows > 0) return BIS_OK;
        A->sw8_state = -1;
    }
}

static bis_status w8_try_rows(bis_ctx *ctx, bis_mat *A, int R, bool *window_too_large) {
    *window_too_large = false;
    const int64_t nb64 = (A->n_rows + (int64_t)kSwRows * R - 1) / ((int64_t)kSwRows * R);
    if (nb64 > (int64_t)1 << 26) return BIS_OK;
    const int nb = (int)nb64;
    bis_sellwin *sw = new bis_sellwin;
    A->sw8 = sw;
    sw->n_blocks = nb;
    sw->R = R;
    sw->n_slices = (int64_t)nb * 4 * R;
    sw->fmt = 5;
    int32_t *slice_chunks = nullptr;
    void *tmp = nullptr;
    int *status = (int *)ctx->counters + 52;
    W8_CHECK(hipMalloc(&sw->hdr, sizeof(int32_t) * 2 * kW8Runs * (size_t)nb));
    W8_CHECK(hipMalloc(&sw->own_rank, sizeof(int32_t) * (size_t)nb));
    W8_CHECK(hipMemsetAsync(sw->own_rank, 0xFF, sizeof(int32_t) * (size_t)nb, ctx->stream)); // (blocks the plan leaves early: -1)
    W8_CHECK(hipMalloc(&sw->row_of, sizeof(uint16_t) * (size_t)nb * kSwRows * (size_t)R));
    W8_CHECK(hipMalloc(&slice_chunks, sizeof(int32_t) * (size_t)(sw->n_slices + 1)));
    W8_CHECK(hipMalloc(&sw->slice_chunk0, sizeof(int64_t) * (size_t)(sw->n_slices + 1)));
    W8_CHECK(hipMemsetAsync(status, 0, 2 * sizeof(int), ctx->stream));
    W8_CHECK(hipMemsetAsync(status + 5, 0, sizeof(int), ctx->stream));
    W8_CHECK(hipMemsetAsync(slice_chunks, 0, sizeof(int32_t) * (size_t)(sw->n_slices + 1), ctx->stream));
    W8_CHECK(hipMemsetAsync(sw->hdr, 0, sizeof(int32_t) * 2 * kW8Runs * (size_t)nb, ctx->stream));
    if (A->rp64) hipLaunchKernelGGL(w8_plan_kernel<int64_t>, dim3(nb), dim3(256), 0, ctx->stream, (const int64_t *)A->row_ptr, A->col, A->n_rows, R, kSwMaxGran, A->view_row0, sw->hdr, slice_chunks, sw->own_rank, sw->row_of, status);
    else hipLaunchKernelGGL(w8_plan_kernel<int32_t>, dim3(nb), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col, A->n_rows, R, kSwMaxGran, A->view_row0, sw->hdr, slice_chunks, sw->own_rank, sw->row_of, status);
    W8_CHECK(hipGetLastError());
    int h[2] = {0, 0};
    W8_CHECK(hipMemcpyAsync(h, status, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    const int64_t ns1 = sw->n_slices + 1;
    hipLaunchKernelGGL(sw_widen_kernel, dim3((unsigned)((ns1 + 255) / 256)), dim3(256), 0, ctx->stream, slice_chunks, sw->slice_chunk0, ns1);
    W8_CHECK(hipGetLastError());
    size_t tmp_bytes = 0;
    W8_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, sw->slice_chunk0, sw->slice_chunk0, (int64_t)0, (size_t)ns1, rocprim::plus<int64_t>(), ctx->stream));
    W8_CHECK(hipMalloc(&tmp, tmp_bytes));
    W8_CHECK(rocprim::exclusive_scan(tmp, tmp_bytes, sw->slice_chunk0, sw->slice_chunk0, (int64_t)0, (size_t)ns1, rocprim::plus<int64_t>(), ctx->stream));
    int64_t total = 0;
    W8_CHECK(hipMemcpyAsync(&total, sw->slice_chunk0 + sw->n_slices, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
    W8_CHECK(hipStreamSynchronize(ctx->stream));
    hipFree(tmp); tmp = nullptr;
    hipFree(slice_chunks); slice_chunks = nullptr;
    sw->total_chunks = total;
    sw->max_gran = h[1];
    // not representable (more than 64 runs / 60 KiB of window in some block), or more than 12 % of padding (every padded entry costs
    // 10 streamed bytes: Anderson's 7-entry rows in 8 slots, 14 %, run 0.310 ms against 0.297 for the row-block kernel): that kernel stays
    if (h[0] || (double)total * 256.0 > 1.12 * (double)A->nnz + 256.0 * 4 * 64) {
        if (getenv("BIS_WIN8_STATS")) fprintf(stderr, "win8 plan (R = %d): %s, %.1f %% padding: not used\n", R, h[0] ? "window not representable" : "representable", 100.0 * ((double)total * 256.0 / (double)A->nnz - 1.0));
        *window_too_large = h[0] != 0;
        bis_spmv_win8_drop(A);
        A->sw8_state = -1;
        return BIS_OK;
    }
    unsigned char *stream = nullptr;
    W8_CHECK(hipMalloc(&stream, (size_t)kW8ChunkBytes * (size_t)(total + 1)));
    sw->codes = reinterpret_cast<uint32_t *>(stream);
    W8_CHECK(hipMemsetAsync(stream + (size_t)total * kW8ChunkBytes, 0, kW8ChunkBytes, ctx->stream));
    if (A->rp64) hipLaunchKernelGGL(w8_fill_kernel<int64_t>, dim3(nb), dim3(256), 0, ctx->stream, (const int64_t *)A->row_ptr, A->col, A->val, A->n_rows, R, sw->hdr, sw->slice_chunk0, sw->row_of, stream);
    else hipLaunchKernelGGL(w8_fill_kernel<int32_t>, dim3(nb), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col, A->val, A->n_rows, R, sw->hdr, sw->slice_chunk0, sw->row_of, stream);
    W8_CHECK(hipGetLastError());
    A->sw8_state = 1;
    if (bis_status tst = w8_tune_placement(ctx, A)) return tst;
    if (getenv("BIS_WIN8_STATS")) fprintf(stderr, "win8 plan (R = %d): %d blocks, window <= %d granules (%zu bytes), %.1f %% padding: used; stream at %p (%zu bytes), hdr %p\n", R, nb, sw->max_gran, (size_t)(2 + 8 * sw->max_gran) * 8, 100.0 * ((double)total * 256.0 / (double)A->nnz - 1.0), (void *)sw->codes, (size_t)kW8ChunkBytes * (size_t)(total + 1), (void *)sw->hdr);
    return BIS_OK;
}
#undef W8_CHECK

bis_status bis_spmv_win8_launch(bis_ctx *ctx, const bis_mat *A, const double *x, double *y, int mode, const double *w,
                                double *partials, const int *stop, int remap_arg, int grid) {
    const bis_sellwin *sw = A->sw8;
    const int x_al16 = ((uintptr_t)x & 15) == 0;
    const size_t lds = (size_t)(2 + 8 * sw->max_gran) * 8;
    const unsigned char *stream = reinterpret_cast<const unsigned char *>(sw->codes);
    const int32_t *own = (mode == 1 && w == x + A->view_row0 && x_al16) ? sw->own_rank : nullptr; // the fused dot's w is x itself (CG: p)
    // chunks requested ahead per lane.  1024-row blocks: 1 where the slices are uniform (HPCG-256 in the CG loop 0.758 ms; 2: 0.771,
    // 3: 0.764; HPCG-512 6.16 / 6.70 / 6.38), 2 with ragged rows (fem:80,80,81: 0.211 ms; 1: 0.221, 3: 0.221); 512-row blocks: 3
    // (HPCG-256 0.861; 2: 0.945, 4: 1.030).  The waves of the other workgroups on the CU cover the rest of the latency.
    const bool ragged = (double)sw->total_chunks * 256.0 > 1.08 * (double)A->nnz;
    const int depth = bis_opts().spmv_win8_depth > 0 ? bis_opts().spmv_win8_depth : (sw->R == 4 ? (ragged ? 2 : 1) : 3);
#define W8_L3(MODE, RR, DD) hipLaunchKernelGGL((spmv_win8_kernel<MODE, RR, DD>), dim3(grid), dim3(256), lds, ctx->stream, x, y, A->n_rows, A->n_cols, \
                                               sw->n_blocks, remap_arg, w, partials, stop, sw->hdr, sw->slice_chunk0, stream, x_al16, own, sw->row_of)
#define W8_L2(MODE, RR) do { if (depth <= 1) W8_L3(MODE, RR, 1); else if (depth == 2) W8_L3(MODE, RR, 2); else if (depth == 3) W8_L3(MODE, RR, 3); else if (depth <= 5) W8_L3(MODE, RR, 4); else W8_L3(MODE, RR, 6); } while (0)
#define W8_L1(MODE) do { if (sw->R == 4) W8_L2(MODE, 4); else if (sw->R == 2) W8_L2(MODE, 2); else W8_L2(MODE, 1); } while (0)
    if (mode == 1) W8_L1(1); else W8_L1(0);
#undef W8_L1
#undef W8_L2
#undef W8_L3
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}
