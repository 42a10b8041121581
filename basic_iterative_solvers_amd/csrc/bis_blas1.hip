// bis_blas1.hip -- context, vectors and the BLAS-1 class kernels of the hot
// path (reference kernels.hpp:119-257, methods/jacobi.hpp:27-40) as
// hand-written gfx950 HIP kernels.  All of them are HBM-bound streaming
// kernels: 16 B per lane (double2) accesses in a grid-stride loop.
//
// What the streams want on MI355X (tools/blas1_bench.hip, N = 16.8 M, operands ROTATING through 8 vectors so that nothing is
// served by the 256 MB Infinity Cache; profiles/r05_b_blas1_variants.log): r = a + s b moves 5.3-5.7 TB/s with plain loads
// and stores, 6.4-6.7 with NON-TEMPORAL LOADS and plain stores, 5.9-6.25 with both non-temporal, 5.5-5.7 with non-temporal
// stores only; two or four independent 16-byte accesses per operand in flight per lane LOSE 3-10 % against one (the waves
// of 8 workgroups per CU already cover the latency; more registers per lane buy nothing), a block-contiguous split instead
// of the grid stride is a wash; grids of 8-16 thousand workgroups gain 2-4 % over 2048.  So: every input stream of the
// elementwise kernels and of the reductions is read with non-temporal loads (a streamed operand does not displace the
// lines the neighbouring kernels of a solver iteration re-use: x-vector tiles of the SpMV, sweep operands), stores stay
// plain (the next kernel of the iteration usually reads what this one wrote), one access per operand per lane and
// iteration, elementwise grids up to 16384 workgroups, reductions up to 8192 (kMaxDotBlocks: their partial sums -- and with
// them the bits of a dot product -- are a function of the grid, so every kernel that reproduces a dot shares that constant).
#include "bis_internal.hpp"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int kEwThreads = 256;
constexpr int kMaxEwBlocks = 16384; // elementwise kernels (no partial sums: the grid is free)

// grid of a reduction (its partials are indexed by block: at most kMaxDotBlocks of them)
inline int ew_grid(int64_t n_items) {
    int64_t g = (n_items + kEwThreads - 1) / kEwThreads;
    if (g < 1) g = 1;
    if (g > kMaxDotBlocks) g = kMaxDotBlocks;
    return (int)g;
}
// grid of an elementwise kernel
inline int ew_grid_wide(int64_t n_items) {
    int64_t g = (n_items + kEwThreads - 1) / kEwThreads;
    if (g < 1) g = 1;
    if (g > kMaxEwBlocks) g = kMaxEwBlocks;
    return (int)g;
}

// streamed (single-touch) input: non-temporal load
typedef double ew_v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld_stream(const double2 *p) {
    const ew_v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const ew_v2d_t *>(p));
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ double ld_stream(const double *p) { return __builtin_nontemporal_load(p); }

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

enum { OP_SUB, OP_SUM, OP_MUL, OP_DIV };

template <int OP>
__device__ __forceinline__ double ew_apply(double a, double b, double s) {
    if (OP == OP_SUB) return fma(-s, b, a);      // a - s*b   kernels.hpp:124
    if (OP == OP_SUM) return fma(s, b, a);       // a + s*b   kernels.hpp:133
    if (OP == OP_MUL) return (a * s) * b;        // a*s*b     kernels.hpp:142
    return a / (s * b);                          // a/(s*b)   kernels.hpp:151
}

// r = op(a, b, s).  r may alias a and/or b (same-index access only).
template <int OP, bool VEC>
__global__ __launch_bounds__(kEwThreads) void ew3_kernel(double *r, const double *a,
                                                         const double *b, int64_t n,
                                                         double s_val, const double *s_dev) {
    const double s = s_dev ? *s_dev : s_val; // device-scalar schedules (GMRES, BiCGSTAB): the factor never visits the host
    const int64_t stride = (int64_t)gridDim.x * kEwThreads;
    int64_t i = (int64_t)blockIdx.x * kEwThreads + threadIdx.x;
    if (VEC) {
        const int64_t n2 = n >> 1;
        const double2 *a2 = reinterpret_cast<const double2 *>(a);
        const double2 *b2 = reinterpret_cast<const double2 *>(b);
        double2 *r2 = reinterpret_cast<double2 *>(r);
        for (; i < n2; i += stride) {
            double2 av = ld_stream(a2 + i), bv = ld_stream(b2 + i), rv;
            rv.x = ew_apply<OP>(av.x, bv.x, s);
            rv.y = ew_apply<OP>(av.y, bv.y, s);
            r2[i] = rv;
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
            r[n - 1] = ew_apply<OP>(a[n - 1], b[n - 1], s);
    } else {
        for (; i < n; i += stride) r[i] = ew_apply<OP>(ld_stream(a + i), ld_stream(b + i), s);
    }
}

enum { U_SCALE, U_COPY, U_FILL };

template <int OP, bool VEC>
__global__ __launch_bounds__(kEwThreads) void ew2_kernel(double *r, const double *a,
                                                         int64_t n, double s_val, const double *s_dev) {
    const double s = s_dev ? *s_dev : s_val;
    const int64_t stride = (int64_t)gridDim.x * kEwThreads;
    int64_t i = (int64_t)blockIdx.x * kEwThreads + threadIdx.x;
    if (VEC) {
        const int64_t n2 = n >> 1;
        const double2 *a2 = reinterpret_cast<const double2 *>(a);
        double2 *r2 = reinterpret_cast<double2 *>(r);
        for (; i < n2; i += stride) {
            double2 rv;
            if (OP == U_FILL) {
                rv.x = s; rv.y = s;
            } else {
                double2 av = ld_stream(a2 + i);
                rv.x = (OP == U_SCALE) ? av.x * s : av.x;
                rv.y = (OP == U_SCALE) ? av.y * s : av.y;
            }
            r2[i] = rv;
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
            r[n - 1] = (OP == U_FILL) ? s : ((OP == U_SCALE) ? a[n - 1] * s : a[n - 1]);
    } else {
        for (; i < n; i += stride)
            r[i] = (OP == U_FILL) ? s : ((OP == U_SCALE) ? a[i] * s : a[i]);
    }
}

// normalize_x, methods/jacobi.hpp:27-40.
__global__ __launch_bounds__(kEwThreads) void normalize_x_kernel(double *x_new,
                                                                 const double *x_old,
                                                                 const double *D,
                                                                 const double *b,
                                                                 int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * kEwThreads;
    for (int64_t i = (int64_t)blockIdx.x * kEwThreads + threadIdx.x; i < n; i += stride) {
        const double d = ld_stream(D + i);
        const double adjusted = fma(-d, ld_stream(x_old + i), ld_stream(x_new + i));
        x_new[i] = (ld_stream(b + i) - adjusted) / d;
    }
}

// out[i] = sum_{k<n_vec} V[k*ldv+i]*y[k], accumulated in k order
// (kernels.hpp:259-271 as called from gmres.hpp:358).
struct MultiAxpyCoef { double y[64]; };
__global__ __launch_bounds__(kEwThreads) void multi_axpy_kernel(const double *V, int64_t ldv,
                                                                MultiAxpyCoef c, int n_vec,
                                                                double *out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * kEwThreads;
    for (int64_t i = (int64_t)blockIdx.x * kEwThreads + threadIdx.x; i < n; i += stride) {
        double acc = 0.0;
        for (int k = 0; k < n_vec; ++k) acc = fma(ld_stream(V + (int64_t)k * ldv + i), c.y[k], acc);
        out[i] = acc;
    }
}

// partials[blockIdx.x] = sum over this block's grid-stride share of a[i]*b[i].  SAME: b is a (a sum of squares: one load per
// element instead of two of the same line; the same products, the same sums).
template <bool VEC, bool SAME>
__global__ __launch_bounds__(kEwThreads) void dot_partial_kernel(const double *a,
                                                                 const double *b, int64_t n,
                                                                 double *partials) {
    __shared__ double lds[kEwThreads / 64];
    const int64_t stride = (int64_t)gridDim.x * kEwThreads;
    int64_t i = (int64_t)blockIdx.x * kEwThreads + threadIdx.x;
    double acc0 = 0.0, acc1 = 0.0;
    if (VEC) {
        const int64_t n2 = n >> 1;
        const double2 *a2 = reinterpret_cast<const double2 *>(a);
        const double2 *b2 = reinterpret_cast<const double2 *>(b);
        for (; i < n2; i += stride) {
            const double2 av = ld_stream(a2 + i), bv = SAME ? av : ld_stream(b2 + i);
            acc0 = fma(av.x, bv.x, acc0);
            acc1 = fma(av.y, bv.y, acc1);
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
            acc0 = fma(a[n - 1], b[n - 1], acc0);
    } else {
        for (; i < n; i += stride) { const double av = ld_stream(a + i); acc0 = fma(av, SAME ? av : ld_stream(b + i), acc0); }
    }
    const double s = block_sum<kEwThreads>(acc0 + acc1, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// Fused step of the modified Gram-Schmidt loop (gmres.hpp:10-27): w -= s * u, then partials of (w, v) over the UPDATED w
// in the same pass (v == nullptr: (w, w), gmres.hpp:36-38).  Index map, accumulators and block sums are those of
// ew3_kernel<OP_SUB> followed by dot_partial_kernel: same bits as the two separate launches.
template <bool VEC>
__global__ __launch_bounds__(kEwThreads) void axpy_dot_kernel(double *w, const double *u, const double *v, int64_t n,
                                                              const double *s_dev, double *partials) {
    __shared__ double lds[kEwThreads / 64];
    const double s = *s_dev;
    const int64_t stride = (int64_t)gridDim.x * kEwThreads;
    int64_t i = (int64_t)blockIdx.x * kEwThreads + threadIdx.x;
    double acc0 = 0.0, acc1 = 0.0;
    if (VEC) {
        const int64_t n2 = n >> 1;
        double2 *w2 = reinterpret_cast<double2 *>(w);
        const double2 *u2 = reinterpret_cast<const double2 *>(u);
        const double2 *v2 = reinterpret_cast<const double2 *>(v);
        for (; i < n2; i += stride) {
            double2 wv = ld_stream(w2 + i);
            const double2 uv = ld_stream(u2 + i);
            wv.x = ew_apply<OP_SUB>(wv.x, uv.x, s);
            wv.y = ew_apply<OP_SUB>(wv.y, uv.y, s);
            w2[i] = wv;
            const double2 vv = v ? ld_stream(v2 + i) : wv;
            acc0 = fma(wv.x, vv.x, acc0);
            acc1 = fma(wv.y, vv.y, acc1);
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const double t = ew_apply<OP_SUB>(w[n - 1], u[n - 1], s);
            w[n - 1] = t;
            acc0 = fma(t, v ? v[n - 1] : t, acc0);
        }
    } else {
        for (; i < n; i += stride) {
            const double t = ew_apply<OP_SUB>(w[i], u[i], s);
            w[i] = t;
            acc0 = fma(t, v ? v[i] : t, acc0);
        }
    }
    const double r = block_sum<kEwThreads>(acc0 + acc1, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// result[v] = sum_i partials[v*stride + i], fixed order.
__global__ __launch_bounds__(256) void reduce_finish_kernel(const double *partials,
                                                            int n_partials, size_t stride,
                                                            double *result) {
    __shared__ double lds[4];
    const double *p = partials + (size_t)blockIdx.x * stride;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_partials; i += 256) acc += p[i];
    const double s = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) result[blockIdx.x] = s;
}

// scalar algebra of the Krylov schedules, one thread: the same IEEE operations, in the same order, the reference's
// host code performs on its doubles
enum { SC_DIV, SC_RATIO_PRODUCT, SC_SQRT_INV };
__global__ void scalar_kernel(int op, double *out, double *out2, const double *a, const double *b, const double *c,
                              const double *d) {
    if (op == SC_DIV) *out = *a / *b;                                    // bicgstab.hpp:34, :51
    else if (op == SC_RATIO_PRODUCT) *out = (*a / *b) * (*c / *d);       // bicgstab.hpp:71
    else { const double nrm = sqrt(*a); *out = nrm; *out2 = 1.0 / nrm; } // kernels.hpp:202, gmres.hpp:44-46
}

} // namespace

bis_status bis_reduce_finish(bis_ctx *ctx, int n_partials, int n_values, size_t stride,
                             double *result_dev) {
    hipLaunchKernelGGL(reduce_finish_kernel, dim3(n_values), dim3(256), 0, ctx->stream,
                       ctx->partials, n_partials, stride, result_dev);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}

bis_status bis_ensure_partials(bis_ctx *ctx, size_t n) {
    if (n <= ctx->partials_cap) return BIS_OK;
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->partials) BIS_HIP_CHECK(ctx, hipFree(ctx->partials));
    ctx->partials = nullptr;
    ctx->partials_cap = 0;
    BIS_HIP_CHECK(ctx, hipMalloc(&ctx->partials, sizeof(double) * n));
    ctx->partials_cap = n;
    return BIS_OK;
}

bis_status bis_fault_check(bis_ctx *ctx) {
    volatile unsigned *f = ctx->fault_host;
    if (!f || *f == 0) return BIS_OK;
    *f = 0; // reported once
    ctx->err = "a triangular sweep lost a hand-off: a row waited for a result that was never published "
               "(a workgroup of the persistent grid could not become resident, or the dependency data is corrupt); "
               "the vectors written since the last blocking call are invalid";
    return BIS_ERR_SYNC;
}

// ---- options: ONE table drives the environment, bis_set_option and bis_options_describe ---------------------------
// X(name): settable through bis_set_option("name", v) AND, at first use, the environment variable BIS_<NAME>;
// Y(name): bis_set_option only (per-call tuning knobs and test hooks: nothing a stray environment should reach).
#define BIS_OPTIONS_ENV(X) X(spmv_variant) X(spmv_window) X(spmv_chunk) X(spmv_chunk_fused) X(spmv_xcd_remap) X(trsv_grid) X(ilu0_wgs) X(trsv_wave_wgs) X(trsv_trial) X(trsv_one_xcd) X(trsv_host_analysis) X(ilu0_wave) X(spmv_packed) X(spmv_packed32) X(spmv_valdict) X(spmv_sellwin) X(device_share) X(spmv_sellwin_rows) X(spmv_sellwin_joint) X(spmv_sellwin_pairs) X(spmv_sellwin_masks) X(spmv_colslab) X(spmv_win8) X(spmv_win8_rows) X(spmv_win8_depth) X(spmv_win8_tune) X(grid_autodetect) X(tune_placement) X(cg_graph) X(force_rp64) X(trsv_tiled) X(trsv_chain) X(trsv_tile_rows) X(trsv_tile_wgs) X(trsv_tile_edge) X(trsv_tile_backoff)
#define BIS_OPTIONS_API(Y) Y(trsv_batch) Y(trsv_wave) Y(ilu0_persistent) Y(trsv_by_pos) Y(spmv_lds_pad) Y(trsv_chain_idle) Y(trsv_chain_pause) Y(trsv_chain_pairs) Y(trsv_chain_prefix) Y(trsv_tile_exp) Y(cg_nt_x) Y(spmv_sellwin_nt) Y(dist_host_plan) Y(trsv_inject_loss) Y(trsv_inject_oom)

namespace {
std::string &opts_env_seen() { static std::string s; return s; } // "BIS_X=v BIS_Y=w": the variables found at first use
std::string upper_env(const char *name) {
    std::string e = "BIS_";
    for (const char *c = name; *c; ++c) e += (char)((*c >= 'a' && *c <= 'z') ? *c - 32 : *c);
    return e;
}
} // namespace

bis_options &bis_opts() {
    static bis_options o = [] {
        bis_options v;
#define X(name) { const std::string en = upper_env(#name); if (const char *e = getenv(en.c_str())) { v.name = atoi(e); \
                  std::string &seen = opts_env_seen(); if (!seen.empty()) seen += ' '; seen += en + "=" + e; } }
        BIS_OPTIONS_ENV(X)
#undef X
        return v;
    }();
    return o;
}

// ---- context ------------------------------------------------------------------
extern "C" {

bis_status bis_set_option(const char *name, int value) {
    if (!name) return BIS_ERR_INVALID;
    bis_options &o = bis_opts();
#define X(opt) if (!strcmp(name, #opt)) { o.opt = value; return BIS_OK; }
    BIS_OPTIONS_ENV(X)
    BIS_OPTIONS_API(X)
#undef X
    return BIS_ERR_INVALID;
}

// The options in effect, for bench / CLI records: a JSON object {"option": value, ..., "env": "BIS_X=v ..."} of every option
// that is not at its default (-1) and the BIS_* variables this process found in its environment at first use.  Returns the
// length needed (excluding the terminator); writes at most cap - 1 characters.
int bis_options_describe(char *buf, int cap) {
    const bis_options &o = bis_opts();
    std::string s = "{";
    bool first = true;
#define X(opt) if (o.opt != -1) { if (!first) s += ", "; first = false; s += std::string("\"") + #opt + "\": " + std::to_string(o.opt); }
    BIS_OPTIONS_ENV(X)
    BIS_OPTIONS_API(X)
#undef X
    if (!first) s += ", ";
    s += "\"env\": \"";
    for (char c : opts_env_seen()) if (c != '"' && c != '\\' && (unsigned char)c >= 32) s += c;
    s += "\"}";
    if (buf && cap > 0) { const int n = std::min((int)s.size(), cap - 1); memcpy(buf, s.data(), (size_t)n); buf[n] = 0; }
    return (int)s.size();
}

int bis_abi_version(void) { return 1; }

bis_status bis_ctx_create(int device, void *stream, bis_ctx **out) {
    if (!out) return BIS_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return BIS_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return BIS_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return BIS_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return BIS_ERR_NO_DEVICE;
    bis_ctx *ctx = new bis_ctx;
    ctx->device = device;
    ctx->n_cus = prop.multiProcessorCount;
    ctx->arch = prop.gcnArchName;
    ctx->hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (ctx->arch.rfind("gfx950", 0) != 0) {
        // the code object in this library is gfx950-only; refuse loudly
        fprintf(stderr, "bis_hip: device %d is %s, this library is built for gfx950 only\n",
                device, ctx->arch.c_str());
        delete ctx;
        return BIS_ERR_NO_DEVICE;
    }
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return BIS_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    ctx->partials_cap = std::max((size_t)kMaxReduceBlocks * 4, (size_t)kMaxDotBlocks * 2);
    bool ok = hipMalloc(&ctx->partials, sizeof(double) * ctx->partials_cap) == hipSuccess &&
              hipMalloc(&ctx->scalars_dev, sizeof(double) * 64) == hipSuccess &&
              hipHostMalloc(&ctx->scalars_host, sizeof(double) * 64) == hipSuccess &&
              hipMalloc(&ctx->counters, sizeof(unsigned) * 64) == hipSuccess &&
              hipMemset(ctx->counters, 0, sizeof(unsigned) * 64) == hipSuccess &&
              hipMemset(ctx->scalars_dev, 0, sizeof(double) * 64) == hipSuccess &&
              hipHostMalloc(&ctx->fault_host, sizeof(unsigned) * 16, hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer((void **)&ctx->fault_dev, ctx->fault_host, 0) == hipSuccess;
    if (ok) memset(ctx->fault_host, 0, sizeof(unsigned) * 16);
    if (!ok) {
        delete ctx;
        return BIS_ERR_HIP;
    }
    *out = ctx;
    return BIS_OK;
}

bis_status bis_ctx_destroy(bis_ctx *ctx) {
    BIS_CTX_OK(ctx);
    hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->prof_events) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (auto &p : ctx->prof_sweep_events) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    hipFree(ctx->partials);
    hipFree(ctx->scalars_dev);
    hipHostFree(ctx->scalars_host);
    hipFree(ctx->counters);
    hipHostFree(ctx->fault_host);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return BIS_OK;
}

const char *bis_last_error(const bis_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context (no usable gfx950 device)"; }

bis_status bis_sync(bis_ctx *ctx) {
    BIS_CTX_OK(ctx);
    BIS_SYNC_CHECK(ctx);
    return BIS_OK;
}

void *bis_ctx_stream(bis_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

bis_status bis_device_info(bis_ctx *ctx, char *arch, size_t arch_len, int *n_cus,
                           int64_t *hbm_bytes) {
    BIS_CTX_OK(ctx);
    if (arch && arch_len) {
        strncpy(arch, ctx->arch.c_str(), arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    if (n_cus) *n_cus = ctx->n_cus;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return BIS_OK;
}

// ---- vectors ------------------------------------------------------------------
bis_status bis_vec_alloc(bis_ctx *ctx, int64_t n, double **out) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, out && n >= 0, "bis_vec_alloc: bad arguments");
    void *p = nullptr;
    BIS_HIP_CHECK(ctx, hipMalloc(&p, sizeof(double) * (size_t)(n > 0 ? n : 1)));
    *out = (double *)p;
    return BIS_OK;
}

bis_status bis_vec_free(bis_ctx *ctx, double *v) {
    BIS_CTX_OK(ctx);
    if (!v) return BIS_OK;
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    BIS_HIP_CHECK(ctx, hipFree(v));
    return BIS_OK;
}

bis_status bis_vec_upload(bis_ctx *ctx, double *dst, const double *src, int64_t n) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, n >= 0 && (n == 0 || (dst && src)), "bis_vec_upload: bad arguments");
    if (n == 0) return BIS_OK;
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyHostToDevice,
                                      ctx->stream));
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return BIS_OK;
}

bis_status bis_vec_download(bis_ctx *ctx, double *dst, const double *src, int64_t n) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, n >= 0 && (n == 0 || (dst && src)), "bis_vec_download: bad arguments");
    if (n == 0) return BIS_OK;
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost,
                                      ctx->stream));
    BIS_SYNC_CHECK(ctx);
    return BIS_OK;
}

} // extern "C"

// ---- elementwise ----------------------------------------------------------------
template <int OP>
static bis_status launch_ew3(bis_ctx *ctx, double *r, const double *a, const double *b,
                             int64_t n, double s, const double *s_dev = nullptr) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, n >= 0 && (n == 0 || (r && a && b)), "elementwise kernel: bad arguments");
    if (n == 0) return BIS_OK;
    const bool vec = aligned16(r) && aligned16(a) && aligned16(b) && n >= 2;
    if (vec)
        hipLaunchKernelGGL((ew3_kernel<OP, true>), dim3(ew_grid_wide(n >> 1)), dim3(kEwThreads), 0,
                           ctx->stream, r, a, b, n, s, s_dev);
    else
        hipLaunchKernelGGL((ew3_kernel<OP, false>), dim3(ew_grid_wide(n)), dim3(kEwThreads), 0,
                           ctx->stream, r, a, b, n, s, s_dev);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}

template <int OP>
static bis_status launch_ew2(bis_ctx *ctx, double *r, const double *a, int64_t n, double s, const double *s_dev = nullptr);

extern "C" {

bis_status bis_subtract_vectors(bis_ctx *ctx, double *r, const double *a, const double *b,
                                int64_t n, double scale) {
    return launch_ew3<OP_SUB>(ctx, r, a, b, n, scale);
}
bis_status bis_sum_vectors(bis_ctx *ctx, double *r, const double *a, const double *b, int64_t n,
                           double scale) {
    return launch_ew3<OP_SUM>(ctx, r, a, b, n, scale);
}
bis_status bis_elemwise_mult_vectors(bis_ctx *ctx, double *r, const double *a, const double *b,
                                     int64_t n, double scale) {
    return launch_ew3<OP_MUL>(ctx, r, a, b, n, scale);
}
bis_status bis_elemwise_div_vectors(bis_ctx *ctx, double *r, const double *a, const double *b,
                                    int64_t n, double scale) {
    return launch_ew3<OP_DIV>(ctx, r, a, b, n, scale);
}
// device-scalar forms: the factor is read from device memory when the kernel runs
bis_status bis_subtract_vectors_dev(bis_ctx *ctx, double *r, const double *a, const double *b,
                                    int64_t n, const double *scale_dev) {
    BIS_REQUIRE(ctx, scale_dev, "bis_subtract_vectors_dev: null scalar");
    return launch_ew3<OP_SUB>(ctx, r, a, b, n, 0.0, scale_dev);
}
bis_status bis_sum_vectors_dev(bis_ctx *ctx, double *r, const double *a, const double *b, int64_t n,
                               const double *scale_dev) {
    BIS_REQUIRE(ctx, scale_dev, "bis_sum_vectors_dev: null scalar");
    return launch_ew3<OP_SUM>(ctx, r, a, b, n, 0.0, scale_dev);
}

} // extern "C"

template <int OP>
static bis_status launch_ew2(bis_ctx *ctx, double *r, const double *a, int64_t n, double s, const double *s_dev) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, n >= 0 && (n == 0 || (r && (a || OP == U_FILL))),
                "elementwise kernel: bad arguments");
    if (n == 0) return BIS_OK;
    const bool vec = aligned16(r) && (OP == U_FILL || aligned16(a)) && n >= 2;
    if (vec)
        hipLaunchKernelGGL((ew2_kernel<OP, true>), dim3(ew_grid_wide(n >> 1)), dim3(kEwThreads), 0,
                           ctx->stream, r, a, n, s, s_dev);
    else
        hipLaunchKernelGGL((ew2_kernel<OP, false>), dim3(ew_grid_wide(n)), dim3(kEwThreads), 0,
                           ctx->stream, r, a, n, s, s_dev);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}

extern "C" {

bis_status bis_scale(bis_ctx *ctx, double *r, const double *v, double scalar, int64_t n) {
    return launch_ew2<U_SCALE>(ctx, r, v, n, scalar);
}
bis_status bis_scale_dev(bis_ctx *ctx, double *r, const double *v, const double *scalar_dev, int64_t n) {
    BIS_REQUIRE(ctx, scalar_dev, "bis_scale_dev: null scalar");
    return launch_ew2<U_SCALE>(ctx, r, v, n, 0.0, scalar_dev);
}
bis_status bis_init_vector(bis_ctx *ctx, double *v, double val, int64_t n) {
    return launch_ew2<U_FILL>(ctx, v, nullptr, n, val);
}
bis_status bis_copy_vector(bis_ctx *ctx, double *out, const double *in, int64_t n) {
    if (out == in) return ctx ? BIS_OK : BIS_ERR_NO_DEVICE;
    return launch_ew2<U_COPY>(ctx, out, in, n, 0.0);
}

bis_status bis_normalize_x(bis_ctx *ctx, double *x_new, const double *x_old, const double *D,
                           const double *b, int64_t n) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, n >= 0 && (n == 0 || (x_new && x_old && D && b)), "bis_normalize_x: bad arguments");
    if (n == 0) return BIS_OK;
    hipLaunchKernelGGL(normalize_x_kernel, dim3(ew_grid_wide(n)), dim3(kEwThreads), 0, ctx->stream,
                       x_new, x_old, D, b, n);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}

bis_status bis_multi_axpy(bis_ctx *ctx, const double *V, int64_t ldv, const double *y_host,
                          int n_vec, double *out, int64_t n) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, n >= 0 && n_vec >= 0 && n_vec <= 64 && (n == 0 || (V && out)) &&
                         (n_vec == 0 || y_host),
                "bis_multi_axpy: bad arguments (n_vec must be <= 64)");
    if (n == 0) return BIS_OK;
    MultiAxpyCoef c;
    for (int k = 0; k < 64; ++k) c.y[k] = k < n_vec ? y_host[k] : 0.0;
    hipLaunchKernelGGL(multi_axpy_kernel, dim3(ew_grid_wide(n)), dim3(kEwThreads), 0, ctx->stream, V,
                       ldv, c, n_vec, out, n);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}

// ---- reductions -------------------------------------------------------------------
bis_status bis_dot_dev(bis_ctx *ctx, const double *a, const double *b, int64_t n,
                       double *result_dev) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, n >= 0 && result_dev && (n == 0 || (a && b)), "bis_dot: bad arguments");
    const bool vec = aligned16(a) && aligned16(b) && n >= 2;
    const int grid = n == 0 ? 1 : (vec ? ew_grid(n >> 1) : ew_grid(n));
#define BIS_DOT_LAUNCH(V, S) hipLaunchKernelGGL((dot_partial_kernel<V, S>), dim3(grid), dim3(kEwThreads), 0, ctx->stream, a, b, n, ctx->partials)
    if (vec) { if (a == b) BIS_DOT_LAUNCH(true, true); else BIS_DOT_LAUNCH(true, false); }
    else { if (a == b) BIS_DOT_LAUNCH(false, true); else BIS_DOT_LAUNCH(false, false); }
#undef BIS_DOT_LAUNCH
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return bis_reduce_finish(ctx, grid, 1, 0, result_dev);
}

bis_status bis_axpy_dot_dev(bis_ctx *ctx, double *w, const double *u, const double *scale_dev, const double *v, int64_t n,
                            double *result_dev) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, n >= 0 && result_dev && scale_dev && (n == 0 || (w && u)), "bis_axpy_dot_dev: bad arguments");
    BIS_REQUIRE(ctx, w != u && w != v, "bis_axpy_dot_dev: w must not alias u or v (pass v = NULL for (w, w))");
    // bit-identical to the separate launches only if both of them would take the same form (16-byte vector or scalar
    // index map); with mixed alignment (odd n: every other basis vector) they would not: run them separately then
    const bool vec_axpy = aligned16(w) && aligned16(u) && n >= 2, vec_dot = aligned16(w) && (!v || aligned16(v)) && n >= 2;
    if (vec_axpy != vec_dot) {
        const bis_status st = bis_subtract_vectors_dev(ctx, w, w, u, n, scale_dev);
        return st != BIS_OK ? st : bis_dot_dev(ctx, w, v ? v : w, n, result_dev);
    }
    const bool vec = vec_axpy;
    const int grid = n == 0 ? 1 : (vec ? ew_grid(n >> 1) : ew_grid(n));
    if (vec) hipLaunchKernelGGL((axpy_dot_kernel<true>), dim3(grid), dim3(kEwThreads), 0, ctx->stream, w, u, v, n, scale_dev, ctx->partials);
    else hipLaunchKernelGGL((axpy_dot_kernel<false>), dim3(grid), dim3(kEwThreads), 0, ctx->stream, w, u, v, n, scale_dev, ctx->partials);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return bis_reduce_finish(ctx, grid, 1, 0, result_dev);
}

bis_status bis_sumsq_dev(bis_ctx *ctx, const double *v, int64_t n, double *result_dev) {
    return bis_dot_dev(ctx, v, v, n, result_dev);
}

bis_status bis_dot(bis_ctx *ctx, const double *a, const double *b, int64_t n,
                   double *result_host) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, result_host, "bis_dot: null result");
    bis_status st = bis_dot_dev(ctx, a, b, n, ctx->scalars_dev);
    if (st != BIS_OK) return st;
    BIS_HIP_CHECK(ctx, hipMemcpyAsync(ctx->scalars_host, ctx->scalars_dev, sizeof(double),
                                      hipMemcpyDeviceToHost, ctx->stream));
    BIS_SYNC_CHECK(ctx);
    *result_host = ctx->scalars_host[0];
    return BIS_OK;
}

bis_status bis_euclidean_vec_norm(bis_ctx *ctx, const double *v, int64_t n,
                                  double *result_host) {
    double ss = 0.0;
    bis_status st = bis_dot(ctx, v, v, n, &ss);
    if (st != BIS_OK) return st;
    *result_host = sqrt(ss); // kernels.hpp:202
    return BIS_OK;
}

bis_status bis_scalar_div(bis_ctx *ctx, double *out_dev, const double *a_dev, const double *b_dev) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, out_dev && a_dev && b_dev, "bis_scalar_div: bad arguments");
    hipLaunchKernelGGL(scalar_kernel, dim3(1), dim3(1), 0, ctx->stream, (int)SC_DIV, out_dev, (double *)nullptr, a_dev, b_dev,
                       (const double *)nullptr, (const double *)nullptr);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}
bis_status bis_scalar_ratio_product(bis_ctx *ctx, double *out_dev, const double *a_dev, const double *b_dev,
                                    const double *c_dev, const double *d_dev) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, out_dev && a_dev && b_dev && c_dev && d_dev, "bis_scalar_ratio_product: bad arguments");
    hipLaunchKernelGGL(scalar_kernel, dim3(1), dim3(1), 0, ctx->stream, (int)SC_RATIO_PRODUCT, out_dev, (double *)nullptr, a_dev, b_dev,
                       c_dev, d_dev);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}
bis_status bis_scalar_sqrt_inv(bis_ctx *ctx, double *norm_dev, double *inv_dev, const double *sumsq_dev) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, norm_dev && inv_dev && sumsq_dev, "bis_scalar_sqrt_inv: bad arguments");
    hipLaunchKernelGGL(scalar_kernel, dim3(1), dim3(1), 0, ctx->stream, (int)SC_SQRT_INV, norm_dev, inv_dev, sumsq_dev,
                       (const double *)nullptr, (const double *)nullptr, (const double *)nullptr);
    BIS_HIP_CHECK(ctx, hipGetLastError());
    return BIS_OK;
}

// ---- profiling ----------------------------------------------------------------------
bis_status bis_profile_enable(bis_ctx *ctx, int on) {
    BIS_CTX_OK(ctx);
    ctx->profile = on != 0;
    return BIS_OK;
}

bis_status bis_profile_read(bis_ctx *ctx, int64_t *spmv_launches, double *spmv_ms) {
    BIS_CTX_OK(ctx);
    BIS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    double ms = 0.0;
    for (size_t i = 0; i < ctx->prof_used; ++i) {
        float t = 0.f;
        BIS_HIP_CHECK(ctx, hipEventElapsedTime(&t, ctx->prof_events[i].first,
                                               ctx->prof_events[i].second));
        ms += t;
    }
    if (spmv_launches) *spmv_launches = (int64_t)ctx->prof_used;
    if (spmv_ms) *spmv_ms = ms;
    ctx->prof_used = 0;
    return BIS_OK;
}

bis_status bis_profile_read_sweeps(bis_ctx *ctx, int64_t *sweeps, double *sweep_ms) {
    BIS_CTX_OK(ctx);
    BIS_SYNC_CHECK(ctx);
    double ms = 0.0;
    for (size_t i = 0; i < ctx->prof_sweep_used; ++i) {
        float t = 0.f;
        BIS_HIP_CHECK(ctx, hipEventElapsedTime(&t, ctx->prof_sweep_events[i].first, ctx->prof_sweep_events[i].second));
        ms += t;
    }
    if (sweeps) *sweeps = (int64_t)ctx->prof_sweep_used;
    if (sweep_ms) *sweep_ms = ms;
    ctx->prof_sweep_used = 0;
    return BIS_OK;
}

} // extern "C"
