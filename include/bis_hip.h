/*
 * bis_hip.h -- C ABI of the MI355X (gfx950) implementation of the
 * SpMV + preconditioner-apply + BLAS-1 hot path of
 * DanecLacey/basic_iterative_solvers.
 *
 * This is the drop-in boundary: every entry point replaces one free function
 * of the reference's operator surface (kernels.hpp / sparse_matrix.hpp /
 * methods/jacobi.hpp) or one step of its accelerator plugin protocol (the
 * SMAX seam: utilities/smax_helpers.hpp, kernels.hpp:44-52).  The reference
 * interface each function replaces is cited as file:line relative to the
 * reference tree.  Signatures are plain C: opaque handles, raw device
 * pointers (`double *` obtained from bis_vec_alloc, exactly where the
 * reference passes `double *` from `new double[N]`), sizes and scalars.
 * No C++/torch types cross this boundary.
 *
 * Conventions
 *   - every function returns a bis_status (0 = BIS_OK); bis_last_error()
 *     gives the message.  The C++ host layer (basic_iterative_solvers_amd/
 *     host/) restores the reference's `void` + exit(EXIT_FAILURE) convention
 *     (common.hpp:382-396).
 *   - one host thread per context; kernels are ordered on the context's HIP
 *     stream; only functions that return a host scalar (bis_dot,
 *     bis_euclidean_vec_norm, downloads, bis_sync) block.
 *   - vector arguments may alias exactly where the reference's callers alias
 *     them: result==operand for the elementwise
 *     kernels, x==b for the triangular solves, output==input for
 *     bis_apply_preconditioner.
 *   - there is NO CPU fallback: if no gfx950 device is usable the context
 *     cannot be created and every entry point fails with BIS_ERR_NO_DEVICE.
 */
#ifndef BIS_HIP_H
#define BIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BIS_API __attribute__((visibility("default")))

typedef int bis_status;
enum {
    BIS_OK = 0,
    BIS_ERR_NO_DEVICE = 1,   /* no usable HIP device / context missing */
    BIS_ERR_INVALID = 2,     /* bad argument (null handle, negative size ...) */
    BIS_ERR_HIP = 3,         /* a HIP runtime call failed */
    BIS_ERR_ZERO_DIAG = 4,   /* SanityChecker::zero_diag, common.hpp:388-391 */
    BIS_ERR_NO_DIAG = 5,     /* SanityChecker::no_diag,   common.hpp:393-396 */
    BIS_ERR_UNSUPPORTED = 6, /* e.g. a row longer than the kernel supports */
    BIS_ERR_COMM = 7,        /* RCCL / halo-exchange failure */
    BIS_ERR_SYNC = 8         /* a device-side wait gave up (lost hand-off in a
                                triangular sweep): results of the work queued
                                since the last blocking call are invalid */
};

/* PrecondType, common.hpp:38-47 (same ordinals). */
enum {
    BIS_PC_NONE = 0,
    BIS_PC_JACOBI = 1,
    BIS_PC_GAUSS_SEIDEL = 2,
    BIS_PC_BACKWARDS_GAUSS_SEIDEL = 3,
    BIS_PC_SYMMETRIC_GAUSS_SEIDEL = 4,
    BIS_PC_TWO_STAGE_GS = 5,
    BIS_PC_SYMMETRIC_TWO_STAGE_GS = 6,
    BIS_PC_ILU0 = 7
};

typedef struct bis_ctx bis_ctx; /* device + stream + scratch (SMAX::Interface
                                   role, preprocessing.hpp:52-65) */
typedef struct bis_mat bis_mat; /* device-resident MatrixCRS,
                                   sparse_matrix.hpp:59-179 */

/* ---- context ------------------------------------------------------------ */
/* `stream` is a hipStream_t to run on (NULL: the context creates its own). */
BIS_API bis_status bis_ctx_create(int device, void *stream, bis_ctx **out);
BIS_API bis_status bis_ctx_destroy(bis_ctx *ctx);
BIS_API const char *bis_last_error(const bis_ctx *ctx);
BIS_API bis_status bis_sync(bis_ctx *ctx);
BIS_API void *bis_ctx_stream(bis_ctx *ctx);
/* "gfx950", CU count, HBM bytes -- for logs and the bench JSON. */
BIS_API bis_status bis_device_info(bis_ctx *ctx, char *arch, size_t arch_len,
                                   int *n_cus, int64_t *hbm_bytes);
/* Tuning knobs, process-wide: "spmv_variant", "spmv_window", "spmv_chunk",
 * "spmv_valdict" (0: no value dictionary, see bis_mat_spmv_stream_info),
 * "trsv_grid" (-1 = default), "force_rp64" (1: 64-bit row pointers at any
 * size), "trsv_tiled" (natural-order triangular sweeps: -1 = the tiled sweep
 * where its plan can be built on the device, i.e. on matrices with a grid
 * hint; 1 = also elsewhere, with the host-built plan; 0 = level-scheduled
 * kernels only; 2 = host-built plan only, for tests).  Matrices created
 * afterwards pick them up. */
BIS_API bis_status bis_set_option(const char *name, int value);
/* The options in effect, for bench / CLI records (no reference counterpart:
 * the reference fixes its configuration at compile time, CMakeLists.txt:19-29):
 * a JSON object of every option that is not at its default, plus "env": the
 * BIS_* environment variables this process found at first use.  Returns the
 * length needed; writes at most cap - 1 characters and a terminator. */
BIS_API int bis_options_describe(char *buf, int cap);
/* number of exported kernel-level symbols, for the load test */
BIS_API int bis_abi_version(void);

/* ---- vectors: replaces `new double[N]` / delete[] in Solver::
 * allocate_structs (solver.hpp:82-110, :130-145) --------------------------- */
BIS_API bis_status bis_vec_alloc(bis_ctx *ctx, int64_t n, double **out);
BIS_API bis_status bis_vec_free(bis_ctx *ctx, double *v);
BIS_API bis_status bis_vec_upload(bis_ctx *ctx, double *dst_dev,
                                  const double *src_host, int64_t n);
BIS_API bis_status bis_vec_download(bis_ctx *ctx, double *dst_host,
                                    const double *src_dev, int64_t n);

/* ---- matrices ------------------------------------------------------------ */
/* MatrixCRS(n_rows, n_cols, nnz) + array fill (sparse_matrix.hpp:76-89) and
 * SMAX register_A (smax_helpers.hpp:10-11): uploads host CRS arrays exactly
 * as given (int32 row_ptr/col, fp64 val, arbitrary column order inside a
 * row) and builds the row-block metadata the SpMV kernel uses.  The structure
 * is checked on the device (row_ptr monotone from 0 to nnz, columns inside
 * [0, n_cols)): BIS_ERR_INVALID instead of a memory fault later. */
BIS_API bis_status bis_mat_create(bis_ctx *ctx, int64_t n_rows, int64_t n_cols,
                                  int64_t nnz, const int32_t *row_ptr,
                                  const int32_t *col, const double *val,
                                  bis_mat **out);
/* Same with 64-bit row pointers (nnz >= 2^31, e.g. HPCG-512: SURVEY.md
 * section 5 defect 6 -- not representable in the reference's MatrixCRS). */
BIS_API bis_status bis_mat_create64(bis_ctx *ctx, int64_t n_rows,
                                    int64_t n_cols, int64_t nnz,
                                    const int64_t *row_ptr, const int32_t *col,
                                    const double *val, bis_mat **out);
BIS_API bis_status bis_mat_destroy(bis_ctx *ctx, bis_mat *A);
BIS_API bis_status bis_mat_info(const bis_mat *A, int64_t *n_rows,
                                int64_t *n_cols, int64_t *nnz);
/* bytes per row pointer on the device: 4, or 8 when nnz >= 2^31 (HPCG-512) or
 * bis_set_option("force_rp64", 1) was in effect when the matrix was made (the
 * tests run the 64-bit instantiations at small sizes that way). */
BIS_API int bis_mat_rp_width(const bis_mat *A);
/* What the SpMV streams per non-zero for this matrix (decided, and built, at
 * the first call of this function or of bis_spmv): col_bytes 4 (CRS columns) or
 * 2 (packed 16-bit column codes); val_bytes 8 (CRS values) or 1 (value
 * dictionary: the matrix has n_dict <= 256 distinct values, compared bit for
 * bit -- every constant-coefficient stencil -- which the kernel keeps in LDS and
 * indexes with a 1-byte code per non-zero; n_dict = 0 without a dictionary);
 * form 0 = the CRS-value kernel, 1 = dictionary kernel, consecutive non-zeros
 * per lane, 2 = dictionary kernel, a lane per row with the codes staged through
 * LDS (rows of at most 40 entries, at most 8 column windows per 256 rows), 3 =
 * the same with the diagonal entries' values in a per-row array beside the
 * dictionary (matrices whose off-diagonal values are few but whose diagonal is
 * not, e.g. the Anderson model: +8 bytes per row), 4 / 5 = forms 2 / 3 with the
 * block's x entries copied into an LDS window by coalesced loads and the codes
 * stored per 64-row slice in lane order (sliced ELL, 12 bytes per 4 non-zeros
 * and lane, short rows padded with an arithmetically neutral entry; option
 * "spmv_sellwin" 0 switches it off).
 * All are lossless re-encodings of the CRS arrays, which stay authoritative:
 * same products, same summation order, bit-identical y.  Option
 * "spmv_valdict" 0 switches the dictionary off, 1 allows form 1 only. */
BIS_API bis_status bis_mat_spmv_stream_info(bis_ctx *ctx, const bis_mat *A,
                                            int *col_bytes, int *val_bytes,
                                            int *n_dict, int *form);
/* Bytes one y = A x launch moves at least with the stream format the matrix
 * currently has (bis_mat_spmv_stream_info): the format's own arrays once -- for
 * forms 4 / 5 including the padding of the sliced-ELL stream --, x once
 * (8 n_cols) and y once (8 n_rows).  The denominator-free part of a roofline
 * figure for the kernel that actually runs; the CRS figure of the reference's
 * loop (kernels.hpp:22-42) is 12 nnz + 20 n_rows. */
BIS_API bis_status bis_mat_spmv_streamed_bytes(bis_ctx *ctx, const bis_mat *A,
                                               int64_t *bytes);
/* Structured-grid hint: the rows are the unknowns of an nx x ny x nz grid, x
 * fastest, dof unknowns per node (row = ((z*ny + y)*nx + x)*dof + d) -- e.g.
 * an HPCG-n.mtx read from a file.  The generators set it themselves, and
 * bis_mat_create recognises one-unknown-per-node stencils of 4096 rows or more
 * from the column offsets of a few rows (option "grid_autodetect" 0: off); strict
 * triangles and ILU(0) factors inherit it.  Only the tiled triangular sweep
 * uses it (tiles that extend in all grid directions); a hint that does not
 * describe the matrix costs speed, never correctness (the tile order is
 * verified against the dependencies, else the natural order is used). */
BIS_API bis_status bis_mat_set_grid_hint(bis_mat *A, int64_t nx, int64_t ny,
                                         int64_t nz, int dof);
/* rebuild a matrix' row-block metadata after bis_set_option (tuning) */
BIS_API bis_status bis_mat_retune(bis_ctx *ctx, bis_mat *A);
/* Placement tuning (setup, optional): WHERE in HBM the streamed arrays of a matrix
 * land moves the SpMV time of the same data by up to 20 % (HPCG-256: 0.82 ... 0.98 ms
 * between allocations of one process).  Re-allocates the
 * streamed arrays (values and column stream) up to max_trials times, times the SpMV
 * on each copy and keeps the fastest; the rejected copies are held until the end so
 * that every trial sees different memory.  Transient memory: up to max_trials copies.
 * Not for row views; call it before views of A are made (bis_dist_create) and
 * before the first triangular solve on A (its plan caches views): refused
 * with BIS_ERR_INVALID afterwards.
 * first_ms / best_ms (optional): SpMV time before and after. */
/* The window + sliced-ELL stream of a matrix (the SpMV's form for matrices without a value
 * dictionary: 8-byte values + 2-byte window slots) is placement-tuned when it is built (streams
 * of 1 GiB or more; option "spmv_win8_tune" = trials, 0 off): what the search did -- re-allocations
 * tried, the kernel's time on the first allocation and on the one kept (ms; zeros when the matrix
 * has no such stream or no tuning ran).  For bench / CLI records. */
BIS_API void bis_mat_win8_tuning(const bis_mat *A, int *trials, double *first_ms, double *kept_ms);
/* Column slabs (the SpMV's form for a matrix WITHOUT locality: rows along which the slab index never falls --
 * ascending columns, the usual case --, a column stream that does not
 * pack, an x that does not fit an XCD's L2 -- config 5's unstructured input as generated): K CRS copies of
 * column ranges whose x slices fit the L2, multiplied in K passes that continue each row's left-to-right sum
 * (kernels.hpp:25-39: the same sum, bit for bit).  Built at the first SpMV; kept where a trial -- three timed
 * launches each way -- measures the passes at least 15 % faster than the one pass (option "spmv_colslab": 0
 * never, k >= 2: k slabs without a trial).  *slabs: K, 0 when the matrix has none; the trial's times (ms,
 * zeros when none ran).  For bench / CLI records. */
BIS_API void bis_mat_colslab_info(const bis_mat *A, int *slabs, double *one_pass_ms, double *slab_passes_ms);
/* debugging / tuning aid (tools/win8_offsets.py): address and size of that stream; set != NULL: the
 * kernel reads the stream from `set` from now on (memory of the caller, into which it has copied the
 * stream), set == NULL: back to the library's own buffer. */
BIS_API bis_status bis_mat_win8_debug_stream(bis_mat *A, void **ptr, size_t *bytes, void *set);
BIS_API bis_status bis_mat_tune_placement(bis_ctx *ctx, bis_mat *A, int max_trials,
                                          double *first_ms, double *best_ms);
/* device addresses of the CRS arrays (tuning / zero-copy interop).  A caller that
 * writes VALUES through them must call bis_mat_retune afterwards: the library
 * caches re-encodings of the values (dictionary code streams, the tiled sweeps'
 * entry streams) that bis_mat_retune and bis_mat_scale_sym drop. */
BIS_API bis_status bis_mat_debug_ptrs(const bis_mat *A, void **row_ptr,
                                      void **col, void **val);
/* copy the device CRS back (tests: bit-exact CRS checks); any pointer may be
 * NULL.  row_ptr is returned as int64. */
BIS_API bis_status bis_mat_download(bis_ctx *ctx, const bis_mat *A,
                                    int64_t *row_ptr, int32_t *col,
                                    double *val);

/* Synthetic inputs generated directly in HBM (stands in
 * for MatrixCOO::scamac_generate, sparse_matrix.hpp:577-721, and for reading
 * HPCG-n.mtx).  Rows [row0,row1) of the global matrix with GLOBAL column
 * indices; n_cols = global row count.
 *   HPCG: 27-point, a_ii=26, a_ij=-1, open boundaries, ascending columns.
 *   Anderson: 7-point periodic L^3, off-diagonals -t, diagonal
 *   W*(u(seed,row)-1/2)+shift, ascending columns. */
BIS_API bis_status bis_mat_gen_hpcg(bis_ctx *ctx, int64_t nx, int64_t ny,
                                    int64_t nz, int64_t row0, int64_t row1,
                                    bis_mat **out);
BIS_API bis_status bis_mat_gen_anderson(bis_ctx *ctx, int64_t L, double t,
                                        double W, double shift, uint64_t seed,
                                        int64_t row0, int64_t row1,
                                        bis_mat **out);

/* FEM-like unstructured input (stands in for reading
 * SuiteSparse Flan_1565.mtx in config 5): nx*ny*nz nodes with 3 unknowns each,
 * 3x3-block couplings to the 27-point neighbours that survive a symmetric coin
 * flip (keep_percent), symmetric negative off-diagonals, strictly dominant
 * diagonal (SPD); ~3*(1+26*keep/100) non-zeros per interior row. */
BIS_API bis_status bis_mat_gen_fem(bis_ctx *ctx, int64_t nx, int64_t ny,
                                   int64_t nz, int keep_percent, uint64_t seed,
                                   int64_t row0, int64_t row1, bis_mat **out);

/* Unstructured input (BASELINE config 5 "SuiteSparse unstructured", read in the
 * reference through sparse_matrix.hpp:225-357; the .mtx is not fetchable): the
 * FEM-like matrix above under a seeded random symmetric permutation of its rows,
 * B = P A P^T, columns ascending inside a row, NO grid hint -- the triangular
 * sweeps and ILU(0) of this matrix take the general (level-scheduled / chunked)
 * kernels like any matrix read from a file.  perm[new] = old: stable ascending
 * order of hash(seed ^ K3, old); written to perm_dev_out (n int32) if non-NULL. */
BIS_API bis_status bis_mat_gen_unstr(bis_ctx *ctx, int64_t nx, int64_t ny,
                                     int64_t nz, int keep_percent, uint64_t seed,
                                     int32_t *perm_dev_out, bis_mat **out);

/* Setup steps kept on the device:
 * split_LU (utilities/LU_factors.hpp:122-309): strict lower / strict upper
 * parts of A, row order preserved; and the diagonal extraction of
 * peel_diag_crs (:827-869): D and 1/D.  Fails with BIS_ERR_ZERO_DIAG /
 * BIS_ERR_NO_DIAG like the reference's SanityChecker. */
BIS_API bis_status bis_mat_split_strict(bis_ctx *ctx, const bis_mat *A,
                                        bis_mat **L_strict, bis_mat **U_strict,
                                        double *D, double *D_inv);

/* -scale on the device: extract_scale
 * (utilities/LU_factors.hpp:880-898), s_r = 1/sqrt(|a_rr|) written to scale[r]
 * (rows without a diagonal entry keep the caller's value), then scale_mat
 * (preprocessing.hpp:15-24), a_rc *= (s_r * s_c), in place.  BIS_ERR_ZERO_DIAG
 * with the reference's message if |a_rr| < 1e-16.  The values only change, so
 * the packed column stream and the row-block tables stay valid. */
BIS_API bis_status bis_mat_scale_sym(bis_ctx *ctx, bis_mat *A, double *scale);

/* Multi-colour symmetric reordering on the device (the
 * role of SMAX's permute_mat, utilities/smax_helpers.hpp:44-80): greedy
 * first-fit colouring in natural row order, rows grouped by colour (stable),
 * B = P A P^T with perm[new] = old written to perm_dev (device, n int32; e.g.
 * storage from bis_vec_alloc).  The strict triangles of B have one dependency
 * level per colour.  BIS_ERR_UNSUPPORTED if more than 64 colours are needed.
 * bis_vec_gather: out[i] = in[perm[i]] (permute b, x_0; out != in). */
BIS_API bis_status bis_mat_multicolour(bis_ctx *ctx, const bis_mat *A,
                                       bis_mat **B, int32_t *perm_dev,
                                       int *n_colours);
BIS_API bis_status bis_vec_gather(bis_ctx *ctx, double *out, const double *in,
                                  const int32_t *perm_dev, int64_t n);
/* Breadth-first (rcm = 0) or reverse Cuthill-McKee (rcm = 1) ordering on the
 * device (SMAX PERM_MODE BFS / RCM roles, CMakeLists.txt:128-133): perm[new] =
 * old written to perm_dev (n int32).  Level-synchronous, and entry for entry
 * the permutation of the sequential queue algorithm (components from the
 * lowest-numbered / lowest-degree unseen vertex, neighbours in ascending index
 * / ascending (degree, index) order).  Structurally symmetric patterns with
 * at most 64 connected components; otherwise BIS_ERR_UNSUPPORTED (the host
 * layer then runs the sequential version).
 * bis_mat_permute: B = P A P^T for any permutation (entries keep their order
 * inside a row, columns renumbered); BIS_ERR_INVALID if perm is not one. */
BIS_API bis_status bis_mat_bfs_order(bis_ctx *ctx, const bis_mat *A, int rcm,
                                     int32_t *perm_dev);
BIS_API bis_status bis_mat_permute(bis_ctx *ctx, const bis_mat *A,
                                   const int32_t *perm_dev, bis_mat **B);
/* out[perm[i]] = in[i]: the inverse permutation without forming it -- returns
 * x* of a permuted solve in the caller's original row order (the reference's
 * SMAX path leaves x* permuted, smax_helpers.hpp:44-80).  out != in. */
BIS_API bis_status bis_vec_scatter(bis_ctx *ctx, double *out, const double *in,
                                   const int32_t *perm_dev, int64_t n);

/* ILU(0) on the device: the arithmetic of the
 * reference's serial factor_ILU0_old (utilities/LU_factors.hpp:320-539),
 * scheduled by the dependency levels of A's strict lower triangle (its
 * level-parallel factor_ILU0_new, :541-768, needs SMAX).  Outputs the strict
 * factors with ascending columns, L_D = 1 and U_D = diag(U); the reference's
 * pivot guard uses ILU0_PIVOT_TOLERANCE / ILU0_PIVOT_REPLACEMENT. */
BIS_API bis_status bis_mat_ilu0(bis_ctx *ctx, const bis_mat *A, double pivot_tol,
                                double pivot_repl, bis_mat **L_strict,
                                bis_mat **U_strict, double *L_D, double *U_D);

/* ---- the operator surface (kernels.hpp) ------------------------------------ */
/* spmv / native_spmv, kernels.hpp:22-52: y = A x. */
BIS_API bis_status bis_spmv(bis_ctx *ctx, const bis_mat *A, const double *x,
                            double *y);
/* sptrsv / native_sptrsv, kernels.hpp:54-86: x = (D + L_strict)^-1 b,
 * natural row order arithmetic; x may alias b. */
BIS_API bis_status bis_sptrsv(bis_ctx *ctx, const bis_mat *L_strict, double *x,
                              const double *D, const double *b);
/* bsptrsv / native_bsptrsv, kernels.hpp:88-117: x = (D + U_strict)^-1 b. */
BIS_API bis_status bis_bsptrsv(bis_ctx *ctx, const bis_mat *U_strict,
                               double *x, const double *D, const double *b);
/* subtract_vectors, kernels.hpp:119-126: r = a - scale*b. */
BIS_API bis_status bis_subtract_vectors(bis_ctx *ctx, double *r,
                                        const double *a, const double *b,
                                        int64_t n, double scale);
/* sum_vectors, kernels.hpp:128-135: r = a + scale*b. */
BIS_API bis_status bis_sum_vectors(bis_ctx *ctx, double *r, const double *a,
                                   const double *b, int64_t n, double scale);
/* elemwise_mult_vectors, kernels.hpp:137-144: r = a*scale*b. */
BIS_API bis_status bis_elemwise_mult_vectors(bis_ctx *ctx, double *r,
                                             const double *a, const double *b,
                                             int64_t n, double scale);
/* elemwise_div_vectors, kernels.hpp:146-153: r = a/(scale*b). */
BIS_API bis_status bis_elemwise_div_vectors(bis_ctx *ctx, double *r,
                                            const double *a, const double *b,
                                            int64_t n, double scale);
/* compute_residual, kernels.hpp:155-162: tmp = A x; res = b - tmp. */
BIS_API bis_status bis_compute_residual(bis_ctx *ctx, const bis_mat *A,
                                        const double *x, const double *b,
                                        double *res, double *tmp);
/* euclidean_vec_norm, kernels.hpp:194-203 (blocking, host result). */
BIS_API bis_status bis_euclidean_vec_norm(bis_ctx *ctx, const double *v,
                                          int64_t n, double *result_host);
/* dot, kernels.hpp:205-212 (blocking, host result). */
BIS_API bis_status bis_dot(bis_ctx *ctx, const double *a, const double *b,
                           int64_t n, double *result_host);
/* stream-ordered variants: the scalar stays on the device (sum of squares,
 * not its root, for the norm) -- for fused / multi-GPU schedules. */
BIS_API bis_status bis_dot_dev(bis_ctx *ctx, const double *a, const double *b,
                               int64_t n, double *result_dev);
BIS_API bis_status bis_sumsq_dev(bis_ctx *ctx, const double *v, int64_t n,
                                 double *result_dev);
/* Device-scalar forms of the axpy-class kernels and the scalar algebra around
 * them: the factor is read from device memory when the kernel runs, so a
 * Krylov iteration needs no host round trip per dot product.  GMRES: the
 * modified Gram-Schmidt step (gmres.hpp:6-53: h = (w,v_j); w -= h v_j; ... ;
 * v_{n+1} = w * (1/||w||)) is j+2 stream-ordered reductions and ONE download
 * of the Hessenberg column instead of j+2 blocking dots; BiCGSTAB
 * (bicgstab.hpp:8-83): alpha, omega, beta stay on the device.  Same kernels,
 * same IEEE operations in the same order as the host-scalar forms: results
 * are bit-identical to them. */
/* One pass for "w -= (*scale_dev) * u; *result_dev = (w, v)" -- the axpy of
 * Gram-Schmidt step j fused with the dot of step j+1 (gmres.hpp:13-14,:25), or,
 * with v = NULL, with the sum of squares of the finished w (:36-38).  Bit-
 * identical to bis_subtract_vectors_dev followed by bis_dot_dev. */
BIS_API bis_status bis_axpy_dot_dev(bis_ctx *ctx, double *w, const double *u,
                                    const double *scale_dev, const double *v,
                                    int64_t n, double *result_dev);
BIS_API bis_status bis_subtract_vectors_dev(bis_ctx *ctx, double *r, const double *a,
                                            const double *b, int64_t n,
                                            const double *scale_dev);
BIS_API bis_status bis_sum_vectors_dev(bis_ctx *ctx, double *r, const double *a,
                                       const double *b, int64_t n,
                                       const double *scale_dev);
BIS_API bis_status bis_scale_dev(bis_ctx *ctx, double *r, const double *v,
                                 const double *scalar_dev, int64_t n);
/* out = a / b  (alpha, omega: bicgstab.hpp:34, :51) */
BIS_API bis_status bis_scalar_div(bis_ctx *ctx, double *out_dev, const double *a_dev,
                                  const double *b_dev);
/* out = (a / b) * (c / d)  (beta: bicgstab.hpp:71) */
BIS_API bis_status bis_scalar_ratio_product(bis_ctx *ctx, double *out_dev,
                                            const double *a_dev, const double *b_dev,
                                            const double *c_dev, const double *d_dev);
/* norm = sqrt(sumsq), inv = 1.0 / norm  (kernels.hpp:202, gmres.hpp:44-46) */
BIS_API bis_status bis_scalar_sqrt_inv(bis_ctx *ctx, double *norm_dev, double *inv_dev,
                                       const double *sumsq_dev);
/* scale, kernels.hpp:214-220: r = v*scalar. */
BIS_API bis_status bis_scale(bis_ctx *ctx, double *r, const double *v,
                             double scalar, int64_t n);
/* init_vector, kernels.hpp:236-241. */
BIS_API bis_status bis_init_vector(bis_ctx *ctx, double *v, double val,
                                   int64_t n);
/* copy_vector, kernels.hpp:252-257. */
BIS_API bis_status bis_copy_vector(bis_ctx *ctx, double *out, const double *in,
                                   int64_t n);
/* normalize_x, methods/jacobi.hpp:27-40:
 * x_new = (b - (x_new - D*x_old))/D. */
BIS_API bis_status bis_normalize_x(bis_ctx *ctx, double *x_new,
                                   const double *x_old, const double *D,
                                   const double *b, int64_t n);
/* dgemm_transpose1 as used at gmres.hpp:358 (kernels.hpp:259-271 with
 * n_cols_B = 1): out[i] = sum_{k<n_vec} V[k*ldv+i]*y[k]; y is a HOST array of
 * n_vec (<= 64) coefficients.  (The reference reads y[n_vec] one past the
 * end at a restart; the defined semantics is that term = 0.) */
BIS_API bis_status bis_multi_axpy(bis_ctx *ctx, const double *V, int64_t ldv,
                                  const double *y_host, int n_vec, double *out,
                                  int64_t n);
/* two_stage_gauss_seidel, kernels.hpp:312-333 (inner_iters =
 * PRECOND_INNER_ITERS). */
BIS_API bis_status bis_two_stage_gauss_seidel(bis_ctx *ctx,
                                              const bis_mat *strict,
                                              double *tmp, double *work,
                                              const double *D_inv,
                                              const double *input,
                                              double *output, int64_t n,
                                              int inner_iters);
/* apply_preconditioner, kernels.hpp:336-414: output = M^-1 input
 * (outer_iters = PRECOND_OUTER_ITERS, inner_iters = PRECOND_INNER_ITERS). */
BIS_API bis_status bis_apply_preconditioner(
    bis_ctx *ctx, int precond_type, int64_t n, const bis_mat *L_strict,
    const bis_mat *U_strict, const double *A_D, const double *A_D_inv,
    const double *L_D, const double *U_D, double *output, double *input,
    double *tmp, double *work, int outer_iters, int inner_iters);

/* ---- named kernels: the reference's plugin protocol --------------------------
 * The reference's accelerator seam (SMAX) registers each kernel once under a
 * name with persistent operands, runs it by name, and rebinds operands after
 * the solver's pointer swaps: utilities/smax_helpers.hpp:7-42
 * (register_kernel / register_A / register_B / register_C /
 * set_mat_upper_triang), kernels.hpp:48,82,113 (kernel(name)->run(0, offset,
 * 0)), jacobi.hpp:93 / kernels.hpp:329 (swap_operands), cg.hpp:136-152
 * (args->x->val = ...).  Same protocol here.  SPMV: C = A * B.  SPTRSV: solve
 * (D + A) B = C with A strictly triangular (the reference hands SMAX a
 * triangle with the diagonal inside; here D is a separate vector, as in the
 * native path, registered with bis_kernel_register_D). */
enum { BIS_KERNEL_SPMV = 0, BIS_KERNEL_SPTRSV = 1 };
BIS_API bis_status bis_register_kernel(bis_ctx *ctx, const char *name, int type);
BIS_API bis_status bis_kernel_register_A(bis_ctx *ctx, const char *name,
                                         const bis_mat *A);
BIS_API bis_status bis_kernel_register_B(bis_ctx *ctx, const char *name,
                                         int64_t size, double *vec);
BIS_API bis_status bis_kernel_register_C(bis_ctx *ctx, const char *name,
                                         int64_t size, double *vec);
BIS_API bis_status bis_kernel_register_D(bis_ctx *ctx, const char *name,
                                         const double *diag);
BIS_API bis_status bis_kernel_set_mat_upper_triang(bis_ctx *ctx,
                                                   const char *name, int flag);
/* offsets in elements, as in run(A_offset, B_offset, C_offset); A_offset must
 * be 0 (the reference never passes anything else). */
BIS_API bis_status bis_kernel_run(bis_ctx *ctx, const char *name,
                                  int64_t A_offset, int64_t B_offset,
                                  int64_t C_offset);
BIS_API bis_status bis_kernel_swap_operands(bis_ctx *ctx, const char *name);

/* ---- stationary solvers as device schedules (methods/jacobi.hpp:43-52, :79-107;
 * methods/gauss_seidel.hpp:26-52, :76-105, :119-129) -------------------------
 * Jacobi, Gauss-Seidel and symmetric Gauss-Seidel as SOLVERS: iteration, true
 * residual b - A x of every iterate (record_residual_norm), its norm and
 * check_stopping_criteria (solver.hpp:177-192) run on the device; no host read
 * per iteration.  Jacobi makes ONE SpMV per iteration -- the product A x_k that
 * samples iteration k's residual is the one iteration k+1 starts from -- and
 * fuses residual, norm partials and the step into one pass; the sampled norms are
 * bit-identical to bis_compute_residual + bis_euclidean_vec_norm.  GS / SGS run
 * the reference's operations unchanged, stream-ordered.  After the stop test has
 * fired the remaining enqueued launches are no-ops (SpMVs and sweeps included).
 * x: x_0 on entry; read the result with bis_stat_solution (Jacobi alternates
 * between x and a buffer of its own).  D = diagonal of A; L_strict / U_strict
 * (bis_mat_split_strict) are needed for GS / SGS only. */
enum { BIS_STAT_JACOBI = 0, BIS_STAT_GS = 1, BIS_STAT_SGS = 2 };
typedef struct bis_stat bis_stat;
BIS_API bis_status bis_stat_create(bis_ctx *ctx, int kind, const bis_mat *A,
                                   const bis_mat *L_strict, const bis_mat *U_strict,
                                   const double *D, const double *b, double *x,
                                   bis_stat **out);
/* init_residual: ||b - A x_0|| (returned), stopping threshold tol * that. */
BIS_API bis_status bis_stat_init(bis_ctx *ctx, bis_stat *s, double tol,
                                 double *r0_norm_host);
/* enqueue n_iters iterations (no synchronisation) */
BIS_API bis_status bis_stat_iterate(bis_ctx *ctx, bis_stat *s, int n_iters);
/* blocking: iterations executed, converged flag, residual history [0..iters] */
BIS_API bis_status bis_stat_status(bis_ctx *ctx, bis_stat *s, int *iters,
                                   int *converged, double *hist_host, int hist_cap);
/* blocking: the iterate the last history entry belongs to, copied to x_out */
BIS_API bis_status bis_stat_solution(bis_ctx *ctx, bis_stat *s, double *x_out);
BIS_API bis_status bis_stat_destroy(bis_ctx *ctx, bis_stat *s);

/* ---- fused CG schedule (cg.hpp:6-54 + :162-166, same arithmetic, fewer
 * passes; SURVEY.md section 8d "fused lower bound") -------------------------- */
typedef struct bis_cg bis_cg;
/* Binds the operands of one CG solve: A, optional Jacobi diagonal (NULL: no
 * preconditioner), b, and x (in/out: x_0 on entry).  Owns p, r, z, tmp. */
BIS_API bis_status bis_cg_create(bis_ctx *ctx, const bis_mat *A,
                                 const double *A_D, const double *b, double *x,
                                 bis_cg **out);
BIS_API bis_status bis_cg_destroy(bis_ctx *ctx, bis_cg *cg);
/* General preconditioner for the fused CG (single GPU or distributed handle):
 * z = M^-1 r through bis_apply_preconditioner (kernels.hpp:336-414) with the
 * given operands (all LOCAL to this rank), instead of None / Jacobi.  Pass B
 * then updates r and (r,r) only; the sweep(s) and a stream-ordered (r,z) follow;
 * everything stays on the device as before.  Call before bis_cg_init. */
BIS_API bis_status bis_cg_set_preconditioner(bis_ctx *ctx, bis_cg *cg, int precond_type,
                                             const bis_mat *L_strict, const bis_mat *U_strict,
                                             const double *A_D, const double *A_D_inv,
                                             const double *L_D, const double *U_D,
                                             int outer_iters, int inner_iters);
/* init_residual (cg.hpp:100-118) + init_stopping_criteria (solver.hpp:173):
 * r0 = b - A x0, z0 = M^-1 r0, p0 = z0; returns ||r0||_2 (blocking). */
BIS_API bis_status bis_cg_init(bis_ctx *ctx, bis_cg *cg, double tol,
                               double *r0_norm_host);
/* Runs up to `n_iters` iterations stream-ordered, no host round trip inside:
 * alpha/beta stay on the device, the stopping test of solver.hpp:177-192
 * (converged | NaN/inf) is evaluated on the device after every iteration and
 * later iterations become no-ops once it fires.  Residual norms are appended
 * to the solve's device history.  Non-blocking. */
BIS_API bis_status bis_cg_iterate(bis_ctx *ctx, bis_cg *cg, int n_iters);
/* Blocking: iterations actually performed so far, converged flag, and the
 * residual history (||r_0||..||r_k||, k = iterations) copied to hist_host
 * (capacity hist_cap doubles; may be NULL). */
BIS_API bis_status bis_cg_status(bis_ctx *ctx, bis_cg *cg, int *iters,
                                 int *converged, double *hist_host,
                                 int hist_cap);

/* ---- measurement ------------------------------------------------------------ */
/* HIP-event timing of the kernels launched on the context's stream.  While
 * enabled, each bis_spmv launch (and the SpMV inside bis_cg_iterate) is
 * bracketed by a hipEvent pair; bis_profile_read returns launches and the
 * summed duration in milliseconds and resets the counters (blocking). */
BIS_API bis_status bis_profile_enable(bis_ctx *ctx, int on);
/* ... and every bis_sptrsv / bis_bsptrsv call (all launches of the call, on the
 * context's stream): count and summed milliseconds since the last read.
 * bis_mat_sweep_kernel names the kernel the last sweep of that direction ran on
 * the triangle (static string; "" before the first sweep) -- the timer-tree
 * counterpart of the reference's LIKWID regions `sptrsv` / `backwards-sptrsv`
 * (kernels.hpp:56-58, :90-92). */
BIS_API bis_status bis_profile_read_sweeps(bis_ctx *ctx, int64_t *sweeps, double *sweep_ms);
BIS_API const char *bis_mat_sweep_kernel(const bis_mat *T, int backward);
BIS_API bis_status bis_profile_read(bis_ctx *ctx, int64_t *spmv_launches,
                                    double *spmv_ms);

/* ---- multi-GPU: 1-D row-block partition --------------
 * One process per GPU.  Rank g owns the contiguous global rows
 * [row_starts[g], row_starts[g+1]) of A and the matching slice of every
 * vector.  The reference has no counterpart (single process, OpenMP): this is
 * the layer that replaces its shared-memory loops across devices.  Only two
 * exchange shapes exist: the halo exchange of boundary x entries before an
 * SpMV and the sum all-reduce of 1-2 scalars after a dot / norm. */

/* Host-only planning step (needs no device; unit-tested on CPU).  Input: the
 * local rows in CRS with GLOBAL column indices.  Output: the sorted list of
 * distinct remote columns ("halo", grouped by owner rank because owners hold
 * contiguous row ranges), how many of them each peer owns (recv_counts
 * [n_ranks]), and interior[2] = the longest run [a,b) of local rows that
 * reference no remote column (the part of the SpMV that can overlap the
 * exchange).  halo_cols may be NULL to query n_halo only; halo_cap is its
 * capacity. */
BIS_API bis_status bis_halo_plan(int64_t n_local, const int64_t *row_ptr,
                                 const int32_t *col_global, int n_ranks,
                                 int rank, const int64_t *row_starts,
                                 int64_t *n_halo, int32_t *halo_cols,
                                 int64_t halo_cap, int64_t *recv_counts,
                                 int64_t *interior);

typedef struct bis_dist bis_dist;

/* Diagonal of a row block that still carries GLOBAL column indices (call it
 * before bis_dist_create renumbers them): D[r] = a(r, row_offset + r), the
 * last diagonal entry of a row wins (peel_diag_crs, utilities/LU_factors.hpp:
 * 827-869); D_inv (optional) = 1/D.  BIS_ERR_NO_DIAG / BIS_ERR_ZERO_DIAG with
 * the reference's SanityChecker texts.  This is what gives every rank the
 * Jacobi preconditioner of its rows (config 3: -cg -p j across GPUs). */
BIS_API bis_status bis_mat_diag(bis_ctx *ctx, const bis_mat *A_local,
                                int64_t row_offset, double *D, double *D_inv);
/* The square diagonal block of a row block (entries with row_offset <= col <
 * row_offset + n_rows, columns renumbered to local, order inside a row kept).
 * Triangular sweeps do not cross ranks (a sequential wavefront, SURVEY.md
 * section 8e): a row-partitioned solve preconditions with Gauss-Seidel /
 * SGS / ILU(0) of THIS block on every rank -- block-Jacobi of the sweeps, a
 * different preconditioner from the single-GPU one (its own parity target:
 * tests/dist_worker.py).  Feed it to bis_mat_split_strict / bis_mat_ilu0 and
 * hand the factors to bis_cg_set_preconditioner. */
BIS_API bis_status bis_mat_diag_block(bis_ctx *ctx, const bis_mat *A_local,
                                      int64_t row_offset, bis_mat **block);

/* Transport, provided by the launcher or by bis_dist_use_rccl.  Buffers are
 * DEVICE pointers; the operation must be ordered on `stream` (a hipStream_t).
 * Return 0 on success. */
typedef struct {
    void *user;
    /* in-place sum all-reduce of `count` doubles */
    int (*allreduce_sum)(void *user, void *stream, double *buf, int count);
    /* send send_counts[p] doubles to each peer p from sendbuf (packed in peer
     * order), receive recv_counts[p] doubles from each peer p into recvbuf
     * (packed in peer order) */
    int (*exchange)(void *user, void *stream, const double *sendbuf,
                    const int64_t *send_counts, double *recvbuf,
                    const int64_t *recv_counts, int n_ranks);
} bis_comm_ops;

/* Builds the distributed operator from the local rows (GLOBAL column
 * indices).  A_local is consumed: its columns are renumbered in place to
 * [0,n_local) for owned columns and n_local + k for the k-th halo column, and
 * the handle stays owned by the bis_dist. */
BIS_API bis_status bis_dist_create(bis_ctx *ctx, bis_mat *A_local, int rank,
                                   int n_ranks, const int64_t *row_starts,
                                   bis_dist **out);
BIS_API bis_status bis_dist_destroy(bis_ctx *ctx, bis_dist *d);
/* n_local rows / entries owned, n_ext = n_local + n_halo: every vector that
 * is an SpMV INPUT must be allocated with n_ext entries (the halo tail is
 * filled by the exchange). */
BIS_API bis_status bis_dist_vec_len(const bis_dist *d, int64_t *n_local,
                                    int64_t *n_ext);
/* What this rank needs: the halo columns (global indices, sorted) and the
 * per-owner counts.  The launcher routes each owner its sub-list once at
 * setup (any host-side all-to-all), then calls bis_dist_set_send_lists. */
BIS_API bis_status bis_dist_halo_info(const bis_dist *d, int64_t *n_halo,
                                      int32_t *halo_cols, int64_t halo_cap,
                                      int64_t *recv_counts);
/* What the peers need from this rank: send_counts[p] global column indices
 * per peer p, concatenated in peer order (all inside this rank's row range). */
BIS_API bis_status bis_dist_set_send_lists(bis_ctx *ctx, bis_dist *d,
                                           const int64_t *send_counts,
                                           const int32_t *send_cols_global);
BIS_API bis_status bis_dist_set_comm(bis_ctx *ctx, bis_dist *d,
                                     const bis_comm_ops *ops);
/* RCCL transport over xGMI (ncclSend/ncclRecv pairs for the halo, ncclAllReduce
 * for scalars).  unique_id is the 128-byte ncclUniqueId made by
 * bis_rccl_unique_id on rank 0 and broadcast by the launcher. */
BIS_API bis_status bis_rccl_unique_id(bis_ctx *ctx, void *out128);
BIS_API bis_status bis_dist_use_rccl(bis_ctx *ctx, bis_dist *d,
                                     const void *unique_id128);
/* What the partition costs this rank: halo entries received and boundary
 * entries sent per SpMV (8 bytes each), rows of the interior run that overlaps
 * the exchange, the number of peers it talks to, and the size of the RCCL
 * communicator it joined (ncclCommCount; 0 when the transport is not RCCL). */
BIS_API bis_status bis_dist_stats(const bis_dist *d, int64_t *n_halo,
                                  int64_t *n_send, int64_t *interior_rows,
                                  int *n_neighbours, int *rccl_ranks);
/* bis_mat_spmv_stream_info of this rank's interior rows (the launch that
 * overlaps the halo exchange): which SpMV kernel the partitioned path runs. */
BIS_API bis_status bis_dist_spmv_stream_info(bis_ctx *ctx, const bis_dist *d,
                                             int *col_bytes, int *val_bytes,
                                             int *n_dict, int *form);
/* bis_mat_spmv_streamed_bytes of this rank's distributed SpMV: the stream formats
 * of its three row ranges (boundary, interior, boundary), x = [owned | halo] and
 * y once. */
BIS_API bis_status bis_dist_spmv_streamed_bytes(bis_ctx *ctx, const bis_dist *d,
                                                int64_t *bytes);
/* While bis_profile_enable is on, every halo exchange (on the communication
 * stream) and every scalar all-reduce (on the compute stream) is bracketed by
 * HIP events; returns counts and summed milliseconds and resets (blocking). */
BIS_API bis_status bis_dist_profile_read(bis_ctx *ctx, bis_dist *d,
                                         int64_t *n_exchange, double *exchange_ms,
                                         int64_t *n_allreduce, double *allreduce_ms);
/* y_local = (A x)_local.  x_ext has n_ext entries (owned part filled by the
 * caller); the halo exchange runs on a second stream under the interior rows'
 * SpMV, the boundary rows follow. */
BIS_API bis_status bis_dist_spmv(bis_ctx *ctx, bis_dist *d, double *x_ext,
                                 double *y_local);
/* global dot: local reduction + all-reduce; result_dev (device) always,
 * result_host if non-NULL (blocking). */
BIS_API bis_status bis_dist_dot(bis_ctx *ctx, bis_dist *d, const double *a,
                                const double *b, double *result_dev,
                                double *result_host);
/* Fused CG on the distributed operator: same schedule and handle as bis_cg_*
 * (bis_cg_init / bis_cg_iterate / bis_cg_status / bis_cg_destroy), with the
 * (Ap,p) all-reduce and the batched {(r,z),(r,r)} all-reduce per iteration.
 * b, x, A_D are local slices (n_local entries). */
BIS_API bis_status bis_dist_cg_create(bis_ctx *ctx, bis_dist *d,
                                      const double *A_D, const double *b,
                                      double *x, bis_cg **out);

#ifdef __cplusplus
}
#endif
#endif /* BIS_HIP_H */
