// bis_unstr.hip -- the unstructured input of BASELINE config 5, generated in HBM.
//
// Config 5 names a SuiteSparse matrix (Flan_1565) read through the reference's .mtx reader
// (sparse_matrix.hpp:225-357); the file cannot be fetched here, and the grid-structured `fem:`
// stand-in carries a grid hint that sends its triangular sweeps down the tiled path -- not the path
// a real unstructured mesh takes.  `unstr:` is that stand-in with every trace of the grid removed:
// B = P A P^T for a seeded random permutation of the ROWS, columns ascending inside a row, no hint.
// Same definition as oracle/bis_oracle.c orc_unstr_perm / orc_gen_unstr (bit-identical arrays:
// tests/test_gpu_kernels.py): perm[new] = old is the stable ascending order of the keys
// hash(seed ^ K3, old).
#include "bis_internal.hpp"

#include <rocprim/rocprim.hpp>

namespace {

constexpr unsigned long long kFemK3 = 0x2545F4914F6CDD1Dull;

__global__ __launch_bounds__(256) void unstr_key_kernel(unsigned long long seed, int64_t n, unsigned long long *__restrict__ key,
                                                        int32_t *__restrict__ idx) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        unsigned long long z = (seed ^ kFemK3) * 0x9E3779B97F4A7C15ull + ((unsigned long long)i + 1) * 0xD1B54A32D192ED03ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        key[i] = z ^ (z >> 31);
        idx[i] = (int32_t)i;
    }
}

// one wave per row: every entry is ranked against the whole row (distinct columns) and written to its place
template <typename RP>
__global__ __launch_bounds__(256) void sort_row_entries_kernel(const RP *__restrict__ rp, const int32_t *__restrict__ col,
                                                               const double *__restrict__ val, int64_t n,
                                                               int32_t *__restrict__ ocol, double *__restrict__ oval) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int64_t s = rp[i], e = rp[i + 1];
        for (int64_t a0 = s; a0 < e; a0 += 64) {
            const int64_t pa = a0 + lane;
            const bool act = pa < e;
            const int c = act ? col[pa] : INT32_MAX;
            const double v = act ? val[pa] : 0.0;
            int rank = 0;
            for (int64_t b0 = s; b0 < e; b0 += 64) {
                const int64_t pb = b0 + lane;
                const int cb = pb < e ? col[pb] : INT32_MAX;
                const int cnt = (int)(e - b0 < 64 ? e - b0 : 64);
                for (int j = 0; j < cnt; ++j) {
                    const int o = __shfl(cb, j, 64);
                    rank += (o < c) || (o == c && b0 + j < pa);
                }
            }
            if (act) { ocol[s + rank] = c; oval[s + rank] = v; }
        }
    }
}

} // namespace

extern "C" bis_status bis_mat_gen_unstr(bis_ctx *ctx, int64_t nx, int64_t ny, int64_t nz, int keep_percent, uint64_t seed,
                                        int32_t *perm_dev_out, bis_mat **out) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, out && nx > 0 && ny > 0 && nz > 0, "bis_mat_gen_unstr: bad arguments");
    const int64_t n = 3 * nx * ny * nz;
    BIS_REQUIRE(ctx, n < INT32_MAX, "bis_mat_gen_unstr: too many rows");
    bis_mat *A = nullptr, *B = nullptr, *Cm = nullptr;
    unsigned long long *key = nullptr, *key_out = nullptr;
    int32_t *idx = nullptr, *perm = nullptr;
    void *tmp = nullptr;
    auto done = [&](bis_status rc) {
        hipFree(key); hipFree(key_out); hipFree(idx); hipFree(tmp);
        if (!perm_dev_out) hipFree(perm);
        if (A) bis_mat_destroy(ctx, A);
        if (B) bis_mat_destroy(ctx, B);
        if (rc != BIS_OK && Cm) bis_mat_destroy(ctx, Cm);
        return rc;
    };
#define UN_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->err = std::string("bis_mat_gen_unstr: ") + hipGetErrorString(e_); return done(BIS_ERR_HIP); } } while (0)
    bis_status st = bis_mat_gen_fem(ctx, nx, ny, nz, keep_percent, seed, 0, n, &A);
    if (st != BIS_OK) return done(st);
    UN_CHECK(hipMalloc(&key, 8 * (size_t)n));
    UN_CHECK(hipMalloc(&key_out, 8 * (size_t)n));
    UN_CHECK(hipMalloc(&idx, 4 * (size_t)n));
    if (perm_dev_out) perm = perm_dev_out; else UN_CHECK(hipMalloc(&perm, 4 * (size_t)n));
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->n_cus * 16);
    hipLaunchKernelGGL(unstr_key_kernel, dim3(grid), dim3(256), 0, ctx->stream, (unsigned long long)seed, n, key, idx);
    size_t bytes = 0;
    UN_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, key, key_out, idx, perm, (size_t)n, 0, 64, ctx->stream)); // stable: ties keep the lower index first
    UN_CHECK(hipMalloc(&tmp, std::max<size_t>(bytes, 16)));
    UN_CHECK(rocprim::radix_sort_pairs(tmp, bytes, key, key_out, idx, perm, (size_t)n, 0, 64, ctx->stream));
    st = bis_mat_permute(ctx, A, perm, &B); // rows and columns renumbered, entries in A's order inside a row
    if (st != BIS_OK) return done(st);
    bis_mat_destroy(ctx, A); A = nullptr;
    st = bis_mat_alloc(ctx, n, n, B->nnz, B->rp64, &Cm);
    if (st != BIS_OK) return done(st);
    UN_CHECK(hipMemcpyAsync(Cm->row_ptr, B->row_ptr, (size_t)(n + 1) * (B->rp64 ? 8 : 4), hipMemcpyDeviceToDevice, ctx->stream));
    const unsigned sgrid = (unsigned)std::min<int64_t>((n + 3) / 4, 1 << 20);
    if (B->rp64) hipLaunchKernelGGL(sort_row_entries_kernel<int64_t>, dim3(sgrid), dim3(256), 0, ctx->stream, (const int64_t *)B->row_ptr, B->col, B->val, n, Cm->col, Cm->val);
    else hipLaunchKernelGGL(sort_row_entries_kernel<int32_t>, dim3(sgrid), dim3(256), 0, ctx->stream, (const int32_t *)B->row_ptr, B->col, B->val, n, Cm->col, Cm->val);
    UN_CHECK(hipGetLastError());
    UN_CHECK(hipStreamSynchronize(ctx->stream));
#undef UN_CHECK
    st = bis_mat_finalize(ctx, Cm); // no grid hint: none is set here, and none is guessed (that happens in bis_mat_create only)
    if (st != BIS_OK) return done(st);
    *out = Cm;
    return done(BIS_OK);
}
