// bis_registry.hip -- the reference's plugin protocol (register once, run by
// name, rebind operands): utilities/smax_helpers.hpp:7-42, kernels.hpp:48,82,
// 113,329, cg.hpp:136-152.  Thin dispatch onto bis_spmv / bis_sptrsv /
// bis_bsptrsv.
#include "bis_internal.hpp"

namespace {
bis_named_kernel *find(bis_ctx *ctx, const char *name) {
    if (!name) return nullptr;
    auto it = ctx->kernels.find(name);
    return it == ctx->kernels.end() ? nullptr : &it->second;
}
} // namespace

#define BIS_FIND(k)                                                                   \
    BIS_CTX_OK(ctx);                                                                  \
    bis_named_kernel *k = find(ctx, name);                                            \
    if (!k) { ctx->err = std::string("unknown kernel name: ") + (name ? name : "(null)"); return BIS_ERR_INVALID; }

extern "C" {

bis_status bis_register_kernel(bis_ctx *ctx, const char *name, int type) {
    BIS_CTX_OK(ctx);
    BIS_REQUIRE(ctx, name && (type == BIS_KERNEL_SPMV || type == BIS_KERNEL_SPTRSV),
                "bis_register_kernel: bad arguments");
    bis_named_kernel k;
    k.type = type;
    ctx->kernels[name] = k;
    return BIS_OK;
}

bis_status bis_kernel_register_A(bis_ctx *ctx, const char *name, const bis_mat *A) {
    BIS_FIND(k);
    BIS_REQUIRE(ctx, A, "bis_kernel_register_A: null matrix");
    k->A = A;
    return BIS_OK;
}

bis_status bis_kernel_register_B(bis_ctx *ctx, const char *name, int64_t size, double *vec) {
    BIS_FIND(k);
    k->B = vec;
    k->size_B = size;
    return BIS_OK;
}

bis_status bis_kernel_register_C(bis_ctx *ctx, const char *name, int64_t size, double *vec) {
    BIS_FIND(k);
    k->C = vec;
    k->size_C = size;
    return BIS_OK;
}

bis_status bis_kernel_register_D(bis_ctx *ctx, const char *name, const double *diag) {
    BIS_FIND(k);
    k->D = diag;
    return BIS_OK;
}

bis_status bis_kernel_set_mat_upper_triang(bis_ctx *ctx, const char *name, int flag) {
    BIS_FIND(k);
    k->upper = flag != 0;
    return BIS_OK;
}

bis_status bis_kernel_swap_operands(bis_ctx *ctx, const char *name) {
    BIS_FIND(k);
    std::swap(k->B, k->C);
    std::swap(k->size_B, k->size_C);
    return BIS_OK;
}

bis_status bis_kernel_run(bis_ctx *ctx, const char *name, int64_t A_offset, int64_t B_offset,
                          int64_t C_offset) {
    BIS_FIND(k);
    BIS_REQUIRE(ctx, A_offset == 0, "bis_kernel_run: A_offset must be 0");
    BIS_REQUIRE(ctx, k->A && k->B && k->C, "bis_kernel_run: operands not registered");
    BIS_REQUIRE(ctx, B_offset >= 0 && C_offset >= 0 && B_offset + k->A->n_cols <= k->size_B + (k->type == BIS_KERNEL_SPTRSV ? k->A->n_cols - k->A->n_rows : 0) &&
                         C_offset + k->A->n_rows <= k->size_C,
                "bis_kernel_run: offset out of the registered operand range");
    if (k->type == BIS_KERNEL_SPMV) return bis_spmv(ctx, k->A, k->B + B_offset, k->C + C_offset);
    BIS_REQUIRE(ctx, k->D, "bis_kernel_run: SPTRSV needs its diagonal (bis_kernel_register_D)");
    return k->upper ? bis_bsptrsv(ctx, k->A, k->B + B_offset, k->D, k->C + C_offset)
                    : bis_sptrsv(ctx, k->A, k->B + B_offset, k->D, k->C + C_offset);
}

} // extern "C"
