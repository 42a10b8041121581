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

#include <rocprim/rocprim.hpp>

#include "bis_internal.hpp"

struct bis_sellwin {
    int n_blocks = 0;
    int64_t n_slices = 0, total_chunks = 0;
    int32_t *hdr = nullptr;          // [n_blocks * 64]: words 0..31 first granule of run k, words 32..63 (rank of the run's first granule in the window) | (granules << 16)
    int64_t *slice_chunk0 = nullptr; // [n_slices + 1]
    uint32_t *codes = nullptr;       // [total_chunks * 64 * 3]
    double *dict = nullptr;          // [256] the matrix' dictionary plus the padding value 1.0
    int max_gran = 0;                // largest window of a block, in granules
    bool small = false;              // <= 32 table entries: the value byte holds 8 * index
    bool diag = false;               // one value code stands for the row's own diagonal value (vdiag)
    int pad_idx = 0, diag_idx = 0;
};

namespace {

constexpr int kSwRows = 256;
constexpr int kSwRuns = 32;
constexpr int kSwRunGran = 64; // granules per run at most
constexpr int kSwHash = 4096;
constexpr int kSwMaxGran = 940; // window <= (2 + 8 * 940) * 8 = 60176 bytes; with the tables in front of it < 64 KiB

__device__ __forceinline__ int sw_wave_max(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

// One workgroup per block of 256 rows: the 8-column granules of x the block reads (LDS hash set, bitonic sort), cut
// into runs of consecutive granules; chunks per slice.  status[0] = 1: not representable; status[1] = max granules.
template <typename RP>
__global__ __launch_bounds__(256) void sw_plan_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                      int64_t n_rows, int32_t *__restrict__ hdr,
                                                      int32_t *__restrict__ slice_chunks, int *status) {
    __shared__ int table[kSwHash];
    __shared__ int list[1024];
    __shared__ int cnt, n_runs, failed;
    __shared__ int run_g0[kSwRuns], run_rank[kSwRuns], srt_g0[kSwRuns], srt_rank[kSwRuns + 1];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t r0 = (int64_t)b * kSwRows;
    const int rows = (int)min((int64_t)kSwRows, n_rows - r0);
    for (int i = tid; i < kSwHash; i += 256) table[i] = -1;
    if (tid == 0) { cnt = 0; n_runs = 0; failed = 0; }
    __syncthreads();
    {
        int len = 0;
        if (tid < rows) len = (int)((int64_t)row_ptr[r0 + tid + 1] - (int64_t)row_ptr[r0 + tid]);
        const int m = sw_wave_max(len);
        if ((tid & 63) == 0) slice_chunks[(size_t)b * 4 + (tid >> 6)] = (m + 3) >> 2;
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
                    if (atomicAdd(&cnt, 1) >= kSwMaxGran) atomicExch(&failed, 1);
                    break;
                }
                if (old == g) break;
            }
            h = (h + 1) & (kSwHash - 1);
        }
    }
    __syncthreads();
    const int n = cnt;
    if (failed || n > kSwMaxGran) {
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

__global__ __launch_bounds__(256) void sw_widen_kernel(const int32_t *__restrict__ in, int64_t *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// One workgroup per block: a lane writes the chunks of its row.
template <typename RP, bool SMALL>
__global__ __launch_bounds__(256) void sw_fill_kernel(const RP *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                      const uint8_t *__restrict__ vcode, int64_t vd_base, int64_t n_rows,
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
    const int64_t r = (int64_t)b * kSwRows + tid;
    int64_t rs = 0;
    int len = 0;
    if (r < n_rows) {
        rs = (int64_t)row_ptr[r];
        len = (int)((int64_t)row_ptr[r + 1] - rs);
    }
    const int64_t slice = (int64_t)b * 4 + wv;
    const int64_t c0 = slice_chunk0[slice];
    const int nch = (int)(slice_chunk0[slice + 1] - c0);
    const unsigned pad_byte = SMALL ? (unsigned)pad_idx * 8u : (unsigned)pad_idx;
    for (int c = 0; c < nch; ++c) {
        unsigned cc[4], vv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = 4 * c + q;
            cc[q] = 0;
            vv[q] = pad_byte;
            if (j < len) {
                const int ci = col[rs + j];
                const int g = ci >> 3;
                int lo = 0, hi = nr - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (g0s[mid] <= g) lo = mid; else hi = mid - 1;
                }
                const int slot = 2 + (rk[lo] + (g - g0s[lo])) * 8 + (ci & 7);
                cc[q] = (unsigned)slot * 8u;
                const unsigned vb = vcode[rs + j - vd_base];
                vv[q] = SMALL ? (vb == 255u ? (unsigned)diag_idx * 8u : vb * 8u) : vb;
            }
        }
        uint32_t *dst = codes + ((size_t)(c0 + c) * 64 + lane) * 3;
        dst[0] = cc[0] | (cc[1] << 16);
        dst[1] = cc[2] | (cc[3] << 16);
        dst[2] = vv[0] | (vv[1] << 8) | (vv[2] << 16) | (vv[3] << 24);
    }
}

struct sw_chunk { uint32_t a, b, c; };

template <bool DIAG, bool SMALL>
struct SwLayout {
    static constexpr int kDictBytes = SMALL ? 256 : 2048;
    static constexpr int kDiagOff = kDictBytes;
    static constexpr int kWinOff = kDictBytes + (DIAG ? 2048 : 0);
    static constexpr unsigned kDiagByte = SMALL ? 248u : 255u;
};

// four non-zeros of every row of the wave: acc += dict[value code] * window[column code], in entry order
template <bool DIAG, bool SMALL>
__device__ __forceinline__ void sw_chunk_fma(const unsigned char *lds, const sw_chunk &cd, unsigned diag_rel, double &acc) {
#pragma clang fp contract(off)
    using L = SwLayout<DIAG, SMALL>;
    const unsigned xa[4] = {cd.a & 0xffffu, cd.a >> 16, cd.b & 0xffffu, cd.b >> 16};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned vb = (cd.c >> (8 * q)) & 0xffu;
        unsigned va = SMALL ? vb : vb << 3;
        if (DIAG) va = vb == L::kDiagByte ? diag_rel : va;
        const double xv = *reinterpret_cast<const double *>(lds + L::kWinOff + xa[q]);
        const double v = *reinterpret_cast<const double *>(lds + va);
        const double pr = v * xv;
        acc = acc + pr;
    }
}

template <int Q, bool DIAG, bool SMALL>
__device__ __forceinline__ void sw_consume(const unsigned char *lds, const sw_chunk (&cd)[8], unsigned diag_rel, double &acc) {
#pragma unroll
    for (int q = 0; q < Q; ++q) sw_chunk_fma<DIAG, SMALL>(lds, cd[q], diag_rel, acc);
}

// MODE 0: y = A x.  MODE 1: also partials[4 b + wave] = sum over the wave's rows of y[r] w[r] (CG's (Ap, p)).
template <int MODE, bool DIAG, bool SMALL>
__global__ __launch_bounds__(256) void spmv_sellwin_kernel(
    const double *x, double *__restrict__ y, int64_t n_rows, int64_t n_cols, int n_blocks, int remap_arg, const double *w,
    double *__restrict__ partials, const int *stop, const int32_t *__restrict__ hdr, const int64_t *__restrict__ slice_chunk0,
    const uint32_t *__restrict__ codes, const double *__restrict__ dict_g, const double *__restrict__ vdiag, int x_al16) {
    using L = SwLayout<DIAG, SMALL>;
    if (stop && stop[1]) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int b = remap_arg > 0 ? xcd_remap(blockIdx.x, remap_arg)
                                : (remap_arg < -1 ? xcd_group_remap(blockIdx.x, -remap_arg) : (int)blockIdx.x);
    if (b >= n_blocks) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = hdr[(size_t)b * 64 + lane];
    const int64_t slice = (int64_t)b * 4 + wv;
    const int64_t c0 = slice_chunk0[slice];
    const int nch = (int)(slice_chunk0[slice + 1] - c0);
    const sw_chunk *cp = reinterpret_cast<const sw_chunk *>(codes) + (size_t)c0 * 64 + lane;
    sw_chunk cd[8];
    {
        const int last = max(nch - 1, 0); // (the stream ends with one spare chunk: an empty last slice reads it)
#pragma unroll
        for (int q = 0; q < 8; ++q) cd[q] = cp[(size_t)min(q, last) * 64];
    }
    const int64_t row = (int64_t)b * kSwRows + tid;
    double wr = 0.0;
    if (MODE == 1 && row < n_rows) wr = w[row];
    // tables
    if (tid < L::kDictBytes / 8) reinterpret_cast<double *>(lds)[tid] = dict_g[tid];
    if (DIAG) reinterpret_cast<double *>(lds + L::kDiagOff)[tid] = row < n_rows ? vdiag[row] : 0.0;
    if (tid == 0) *reinterpret_cast<double *>(lds + L::kWinOff) = -0.0;
    // window: the runs of 8-column granules; a run has at most 256 pieces of 16 bytes, a wave takes 64 of them and
    // the hardware writes them to LDS behind the wave-uniform base (no register staging, nothing waited for here)
    {
        const int n_runs = __popcll(__ballot(lane >= 32 && (hw >> 16) != 0));
        unsigned char *win = lds + L::kWinOff + 16;
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
    const unsigned diag_rel = (unsigned)(L::kDiagOff + tid * 8);
    double acc = 0.0;
    for (int cb = 0; cb < nch; cb += 8) {
        const int rem = nch - cb;
        if (cb > 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) cd[q] = cp[(size_t)(cb + min(q, rem - 1)) * 64];
        }
        switch (rem) {
        case 1: sw_consume<1, DIAG, SMALL>(lds, cd, diag_rel, acc); break;
        case 2: sw_consume<2, DIAG, SMALL>(lds, cd, diag_rel, acc); break;
        case 3: sw_consume<3, DIAG, SMALL>(lds, cd, diag_rel, acc); break;
        case 4: sw_consume<4, DIAG, SMALL>(lds, cd, diag_rel, acc); break;
        case 5: sw_consume<5, DIAG, SMALL>(lds, cd, diag_rel, acc); break;
        case 6: sw_consume<6, DIAG, SMALL>(lds, cd, diag_rel, acc); break;
        case 7: sw_consume<7, DIAG, SMALL>(lds, cd, diag_rel, acc); break;
        default: sw_consume<8, DIAG, SMALL>(lds, cd, diag_rel, acc); break;
        }
    }
    if (row < n_rows) y[row] = acc;
    if (MODE == 1) {
        const double t = wave_sum(row < n_rows ? acc * wr : 0.0);
        if (lane == 0) partials[(size_t)b * 4 + wv] = t;
    }
}

bool sw_enabled() { return bis_opts().spmv_sellwin != 0; }

} // namespace

void bis_spmv_sellwin_drop(bis_mat *A) {
    if (A->sw) {
        hipFree(A->sw->hdr); hipFree(A->sw->slice_chunk0); hipFree(A->sw->codes); hipFree(A->sw->dict);
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

// Build the form for a matrix that has a value dictionary (A->vd_state == 1); A->sw_state tells the outcome.
bis_status bis_spmv_sellwin_try(bis_ctx *ctx, bis_mat *A) {
    if (A->sw_state != 0) return BIS_OK;
    A->sw_state = -1;
    if (!sw_enabled() || A->vd_state != 1 || A->n_rows == 0 || A->nnz == 0 || A->n_cols >= ((int64_t)1 << 31) - 16) return BIS_OK;
    const int64_t nb64 = (A->n_rows + kSwRows - 1) / kSwRows;
    if (nb64 > (int64_t)1 << 28) return BIS_OK;
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
    sw->n_slices = (int64_t)nb * 4;
    sw->small = n_tab <= cap_small;
    sw->diag = A->vd_diag;
    sw->pad_idx = pad_idx;
    sw->diag_idx = 31;
    int32_t *slice_chunks = nullptr;
    void *tmp = nullptr;
    int *status = (int *)ctx->counters + 52;
    SW_CHECK(hipMalloc(&sw->hdr, sizeof(int32_t) * 64 * (size_t)nb));
    SW_CHECK(hipMalloc(&slice_chunks, sizeof(int32_t) * (size_t)(sw->n_slices + 1)));
    SW_CHECK(hipMalloc(&sw->slice_chunk0, sizeof(int64_t) * (size_t)(sw->n_slices + 1)));
    SW_CHECK(hipMalloc(&sw->dict, sizeof table));
    SW_CHECK(hipMemcpyAsync(sw->dict, table, sizeof table, hipMemcpyHostToDevice, ctx->stream));
    SW_CHECK(hipMemsetAsync(status, 0, 2 * sizeof(int), ctx->stream));
    SW_CHECK(hipMemsetAsync(slice_chunks + sw->n_slices, 0, sizeof(int32_t), ctx->stream));
    if (A->rp64) hipLaunchKernelGGL(sw_plan_kernel<int64_t>, dim3(nb), dim3(256), 0, ctx->stream, (const int64_t *)A->row_ptr, A->col, A->n_rows, sw->hdr, slice_chunks, status);
    else hipLaunchKernelGGL(sw_plan_kernel<int32_t>, dim3(nb), dim3(256), 0, ctx->stream, (const int32_t *)A->row_ptr, A->col, A->n_rows, sw->hdr, slice_chunks, status);
    SW_CHECK(hipGetLastError());
    int h[2] = {0, 0};
    SW_CHECK(hipMemcpyAsync(h, status, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
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
    sw->total_chunks = total;
    sw->max_gran = h[1];
    // not representable, or more than 30 % of padding (ragged rows): the gather kernels stay
    if (h[0] || (double)total * 256.0 > 1.3 * (double)A->nnz + 256.0 * 4 * 64) {
        bis_spmv_sellwin_drop(A);
        A->sw_state = -1;
        return BIS_OK;
    }
    SW_CHECK(hipMalloc(&sw->codes, sizeof(uint32_t) * 3 * 64 * (size_t)(total + 1)));
    SW_CHECK(hipMemsetAsync(sw->codes + (size_t)total * 192, 0, sizeof(uint32_t) * 192, ctx->stream));
#define SW_FILL(RP, SMALL) hipLaunchKernelGGL((sw_fill_kernel<RP, SMALL>), dim3(nb), dim3(256), 0, ctx->stream, (const RP *)A->row_ptr, A->col, A->vcode, A->vd_base, A->n_rows, sw->hdr, sw->slice_chunk0, sw->codes, sw->pad_idx, sw->diag_idx)
    if (A->rp64) { if (sw->small) SW_FILL(int64_t, true); else SW_FILL(int64_t, false); }
    else { if (sw->small) SW_FILL(int32_t, true); else SW_FILL(int32_t, false); }
#undef SW_FILL
    SW_CHECK(hipGetLastError());
    A->sw_state = 1;
    return BIS_OK;
}

int bis_spmv_sellwin_blocks(const bis_mat *A) { return A->sw_state == 1 ? A->sw->n_blocks : 0; }

// bytes of the form's own arrays one launch reads: the code stream (with its padding), block headers, slice offsets, table
int64_t bis_spmv_sellwin_bytes(const bis_mat *A) {
    if (A->sw_state != 1) return 0;
    return A->sw->total_chunks * 768 + (int64_t)A->sw->n_blocks * 256 + 8 * (A->sw->n_slices + 1) + 2048;
}

// mode 0 / 1 as in the kernel; grid and remap_arg from the caller's block map over bis_spmv_sellwin_blocks(A)
bis_status bis_spmv_sellwin_launch(bis_ctx *ctx, const bis_mat *A, const double *x, double *y, int mode, const double *w,
                                   double *partials, const int *stop, int remap_arg, int grid) {
    const bis_sellwin *sw = A->sw;
    const int x_al16 = ((uintptr_t)x & 15) == 0;
    const size_t win = (size_t)(2 + 8 * sw->max_gran) * 8;
#define SW_L3(MODE, DIAG, SMALL)                                                                                       \
    hipLaunchKernelGGL((spmv_sellwin_kernel<MODE, DIAG, SMALL>), dim3(grid), dim3(256), (SwLayout<DIAG, SMALL>::kWinOff) + win, \
                       ctx->stream, x, y, A->n_rows, A->n_cols, sw->n_blocks, remap_arg, w, partials, stop, sw->hdr,   \
                       sw->slice_chunk0, sw->codes, sw->dict, A->vdiag, x_al16)
#define SW_L2(MODE, DIAG) do { if (sw->small) SW_L3(MODE, DIAG, true); else SW_L3(MODE, DIAG, false); } while (0)
#define SW_L1(MODE) do { if (sw->diag) SW_L2(MODE, true); else SW_L2(MODE, false); } while (0)
    if (mode == 1) SW_L1(1); else SW_L1(0);
#undef SW_L1
#undef SW_L2
#undef SW_L3
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}
