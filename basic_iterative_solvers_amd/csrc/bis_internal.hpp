// bis_internal.hpp -- shared internals of libbis_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "bis_hip.h"

struct bis_trsv_plan;

struct bis_named_kernel {
    int type = 0;
    const bis_mat *A = nullptr;
    double *B = nullptr, *C = nullptr;
    int64_t size_B = 0, size_C = 0;
    const double *D = nullptr;
    bool upper = false;
};

struct bis_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cus = 0;
    std::string arch;
    int64_t hbm_bytes = 0;
    std::string err;

    // reduction scratch: per-block partials + result slots (device), and a
    // pinned host mirror for blocking scalar returns
    double *partials = nullptr;   // [partials_cap], grown on demand
    size_t partials_cap = 0;
    double *scalars_dev = nullptr; // [64]
    double *scalars_host = nullptr; // pinned [64]
    unsigned *counters = nullptr; // [64] tickets / arrival counters (zeroed)
    // device-side waits that gave up (a lost hand-off in a triangular sweep) set this word (pinned host
    // memory mapped into the device); every blocking entry point checks it after its synchronisation
    unsigned *fault_host = nullptr, *fault_dev = nullptr;

    std::map<std::string, bis_named_kernel> kernels; // named-kernel registry (SMAX protocol)

    // set by bis_cg_iterate around its launches: the fused-dot SpMV returns at once when
    // flags[1] (the solver's stop flag) is set, like the other passes of a stopped iteration
    const int *spmv_stop = nullptr;

    // profiling
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_sweep_events; // one pair per bis_sptrsv / bis_bsptrsv call
    size_t prof_sweep_used = 0;
};

// Tuning knobs (bis_set_option / BIS_* environment variables at first use).
struct bis_options;
bis_options &bis_opts();
struct bis_options {
    int spmv_variant = -1; // -1: default (40 = 256 threads, 4 staged vectors)
    int spmv_window = -1;  // -1: default (0: CRS gather kernel)
    int spmv_chunk = -1;   // -1: default (1024 non-zeros + rows per row block)
    int spmv_chunk_fused = -1; // -1: default (2048) for the SpMV with the fused dot epilogue
    int spmv_xcd_remap = -1; // 1: each XCD sweeps its own slab of row blocks (default: blockIdx order)
    int trsv_grid = -1;    // -1: automatic
    int ilu0_wave = -1;    // 0: lane-per-row ILU(0) level kernel (default: wave per row)
    int ilu0_wgs = -1;     // persistent ILU(0): workgroups per CU (default 6; never all 8 the runtime reports: the grid must be resident as a whole)
    int ilu0_persistent = -1; // 0: a launch per level (default: one persistent launch, a flag per finished row)
    int trsv_host_analysis = -1; // 1: level analysis on the host (default: on the device)
    int trsv_wave = -1;    // 1: one wave per row in the sync-free sweeps (default for rows > 16)
    int trsv_wave_wgs = -1; // wave-per-row level sweep: workgroups per CU (default 4; up to the 8 the runtime reports resident)
    int trsv_trial = -1;   // 0: the level-scheduled sweeps never time their two kernels against each other (default: the first sweep of a large triangle does)
    int trsv_batch = -1;   // dependencies polled per round trip (4, 8, 16; -1: by row length)
    int trsv_by_pos = -1;  // 1 (default): sentinel scratch in level order; 0: in row order
    int trsv_one_xcd = -1; // k > 0: sync-free sweeps run on one elected XCD with k workgroups per CU
    int cg_graph = -1;      // 1: bis_cg_iterate replays one captured iteration as a hipGraph (opt-in, measured no gain)
    int tune_placement = -1; // k > 0: bis_dist_create runs bis_mat_tune_placement(A, k) before it makes the row views
    int spmv_packed32 = -1; // 1: also try the 32-window packed format (opt-in)
    int spmv_lds_pad = -1; // diagnostic: extra dynamic LDS bytes per workgroup (lowers occupancy)
    int spmv_packed = -1;  // 16-bit packed column stream: 0 off, 1 select tree, 2 lane permute (-1: default = 1)
    int spmv_valdict = -1; // value dictionary (a matrix with <= 256 distinct values streams 1-byte value codes): 0 off, 1 consecutive form only, 2 lane-per-row form where it applies (-1: default)
    int dist_host_plan = -1; // 1: bis_dist_create plans the halo on the host from the downloaded structure (default: on the device)
    int grid_autodetect = -1; // bis_mat_create: recognise a stencil on a structured grid from the offsets of a few rows (0: off)
    int trsv_chain = -1;    // natural-order sweeps of matrices without a grid: -1 = the chained sweep (bis_trsv_chain.hip) where its plan applies (chains of >= 3 rows on
                            // average, fewer chains straddling a level than resident waves), 1 = also with shorter chains, 0 = level-scheduled kernels only
    int trsv_chain_idle = -1;  // chained sweep: poll rounds without a delivery after which a feeder polls one word only (default: never -- measured slower with the pairs kept near the bound)
    int trsv_chain_pause = -1; // ... and pauses up to this many x 256 cycles between its looks (default 8)
    int trsv_chain_prefix = -1; // 1: the feeder sums the part of a row's fma chain before its first chain-internal operand (default off: measured slower -- the feeder then paces the pair: fem:80,80,81 forward 2.55 -> 3.0 ms)
    int trsv_chain_pairs = -1; // wave pairs per ticket queue (default: the plan's straddle bound + 1/8 + 8)
    int trsv_tiled = -1;    // natural-order sweeps: -1 = tiled sweep (bis_trsv_tiled.hip) where its device plan applies (grid hint), 1 = also with the host plan, 2 = host plan only, 0 = level-scheduled kernels
    int trsv_tile_rows = -1; // rows per tile at most (default 8192 for rows of <= 8 entries, else 2048)
    int trsv_tile_wgs = -1;  // resident workgroups per CU of the tiled sweep (default: what fits, 3)
    int trsv_tile_backoff = -1; // tiled sweep: a poller's pause grows by 64 cycles per round that delivers nothing, up to this many (default 16: HPCG-256 2.35 -> 2.30 ms per sweep, HPCG-128 0.77 -> 0.74, the 7-point grid unchanged; 0: never)
    int spmv_sellwin_nt = -1; // sliced-ELL SpMV: 0 = the code stream through the caches (default: non-temporal loads)
    int cg_nt_x = -1;        // fused CG: 0 = x read and written through the caches in the p-update pass (default: non-temporal)
    int trsv_tile_exp = -1;  // tiled sweep, timing experiments (bis_trsv_tiled.hip, TiledArgs::exp_flags); results are wrong with any bit set
    int trsv_tile_edge = -1; // grid-hinted matrices: tile extents in nodes, e (cubic) or ex | ey << 8 | ez << 16 (default by row length; 0 = interval tiles of the natural order)
    int force_rp64 = -1;   // 1: matrices created afterwards get 64-bit row pointers whatever their size (tests of the HPCG-512 code path)
    int spmv_sellwin = -1; // dictionary SpMV with the block's x window in LDS and sliced-ELL codes (bis_spmv_sell.hip): 0 off (-1: on where the matrix qualifies)
    int spmv_sellwin_rows = -1; // rows per lane of the sliced-ELL form: 1 or 2 (blocks of 256 or 512 rows; -1: default 2)
    int spmv_sellwin_pairs = -1; // 0: never the one-byte (column - row, value) pair codes
    int spmv_sellwin_joint = -1; // 0: never the 16-bit joint (slot, value) codes
    int spmv_colslab = -1;     // column slabs for matrices without locality (bis_spmv_slab.hip): 0 never, -1 where a build-time trial measures them faster, k >= 2: k slabs wherever the plan applies (tests)
    int spmv_win8 = -1;        // window + sliced-ELL SpMV with the 8-byte values streamed (matrices without a dictionary form, or with spmv_valdict = 0): 0 off, 1 on where its plan applies (-1: default = on)
    int spmv_win8_rows = -1;   // ... rows per lane: 1, 2 or 4 (blocks of 256, 512, 1024 rows; default 2)
    int spmv_win8_depth = -1;  // ... chunks requested ahead per lane: 1, 2, 3, 4 or 6 (default by block size)
    int spmv_win8_tune = -1;   // ... placement tuning of the stream at build time: up to k re-allocations, the fastest kept (default 12 for streams of >= 1 GiB; 0 off)
    int spmv_sellwin_masks = -1; // 0: never the per-row pair masks (fmt 4: 4 bytes per ROW where the matrix has at most 32 (column - row, value) pairs)
    int device_share = -1;  // k > 1: this device is shared by k processes that all run persistent grids (several ranks on one GPU in a test
                            // or rehearsal): kernels that need their whole grid resident keep to 1/k of the device
    int trsv_inject_oom = -1;  // test hook: 1 makes the tiled sweep's device plan run out of memory
    int trsv_inject_loss = -1; // test hook: k > 0 makes row k-1 of the next natural-order sweep wait for a result nobody publishes
};
// row_ptr width of a new matrix: int64 when the non-zeros (plus the stream padding) do not fit int32
inline bool bis_want_rp64(int64_t nnz) { return nnz >= (int64_t)INT32_MAX - 8 || bis_opts().force_rp64 > 0; }

constexpr int kMaxReduceBlocks = 2048;
// workgroups (= partial sums) of the stand-alone reductions -- dot, sum of squares, the fused axpy + dot, the Jacobi step's
// residual norm: 16 N bytes of two non-temporal read streams move 5.9 TB/s from a grid of 2048, 6.3 from 4096, 6.4 from 8192
// (tools/blas1_bench.hip).  One constant for all of them: kernels that promise the bits of "ew3 then dot" share the index map.
constexpr int kMaxDotBlocks = 8192;
constexpr int kWinMaxTiles = 128; // x window: at most 128 tiles of 16 columns (16 KiB of LDS)
constexpr int kWinTile = 16;
constexpr int kWinTileLog = 4;

struct bis_mat {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    bool rp64 = false;        // row_ptr width on the device
    bool view = false;        // row-range view: row_ptr/col/val belong to another bis_mat
    void *row_ptr = nullptr;  // int32_t[n_rows+1] or int64_t[n_rows+1]
    int32_t *col = nullptr;   // [nnz + pad]
    double *val = nullptr;    // [nnz + pad]
    // SpMV row-block metadata: block k covers rows [blk_row[k], blk_row[k+1])
    int32_t *blk_row = nullptr;
    int64_t *blk_nnz = nullptr; // row_ptr[blk_row[k]] (saves a dependent load)
    // x-window acceleration structure (bis_spmv.hip): per row block the sorted
    // list of 16-column tiles of x it touches, and per non-zero a 16-bit offset
    // into that window (replaces the 32-bit col in the SpMV stream)
    uint16_t *loc = nullptr;    // [nnz_range + pad], index k - loc_base
    int32_t *tiles = nullptr;   // [n_blocks * kWinMaxTiles]
    int32_t *tile_cnt = nullptr; // [n_blocks]
    int64_t loc_base = 0;
    int max_tiles = 0;
    bool win_ok = false;
    int n_blocks = 0;
    // packed-column streams (bis_spmv.hip), one per row-block table (0 plain, 1
    // fused): per non-zero a 16-bit code (segment:3 | offset:13) against 8 column
    // bases per row block -- 10 instead of 12 streamed bytes per non-zero.  Built
    // lazily at the first SpMV; state 0 = not tried, 1 = usable, -1 = not
    // representable (some block needs more than 8 windows of 8192 columns).
    uint16_t *pk[2] = {nullptr, nullptr};
    int32_t *pk_seg[2] = {nullptr, nullptr}; // [n_blocks * 8]
    int64_t pk_base[2] = {0, 0};
    int pk_state[2] = {0, 0};
    int pk_kind[2] = {0, 0};   // 1: 8 windows x 8192 columns, 3: 32 windows x 2048 columns
    // value dictionary (bis_spmv.hip): a matrix with at most 256 distinct values (compared bit for bit) keeps them in a
    // 256-entry table and streams one byte per non-zero instead of the 8-byte value.  Built lazily at the first SpMV;
    // state 0 = not tried, 1 = usable, -1 = too many distinct values.  The CRS values stay authoritative.
    uint8_t *vcode = nullptr;  // [nnz_range + pad], index k - vd_base
    double *vdict = nullptr;   // [256], ascending bit patterns, unused entries 0
    int64_t vd_base = 0;
    int vd_state = 0, vd_n = 0;
    // ... or all values EXCEPT the diagonal entries (col == row + view_row0) are at most 255: code 255 then stands for "this
    // row's diagonal value", kept in a per-row array (Anderson: random diagonal, constant hopping).  Lane-per-row form only.
    double *vdiag = nullptr;   // [n_rows] when vd_diag
    bool vd_diag = false;
    bool vd_rm_only = false;   // only the lane-per-row kernel can use this dictionary (vd_diag, or the row-block tables have no packed columns)
    int64_t view_row0 = 0;     // row views: first row of the view in the matrix it was cut from
    // lane-per-row form of the dictionary kernel: blocks of 256 rows with their own packed column stream
    int64_t *rm_nnz = nullptr;  // [rm_blocks + 1] row_ptr at every 256th row
    uint16_t *rm_pk = nullptr;  // column codes against rm_seg's windows, index k - rm_base
    int32_t *rm_seg = nullptr;  // [rm_blocks * 8]
    int64_t rm_base = 0;
    int rm_state = 0, rm_blocks = 0, rm_kind = 1; // rm_kind 1: 8 windows x 8192 columns per block, 3: 32 windows x 2048 columns
    // x-window + sliced-ELL form of the dictionary kernel (bis_spmv_sell.hip); state as above
    struct bis_sellwin *sw = nullptr;
    int sw_state = 0;
    // the same plan with the 8-byte values streamed (no dictionary needed): "win8", bis_spmv_sell.hip; state as above
    struct bis_sellwin *sw8 = nullptr;
    std::vector<bis_mat *> *colslabs = nullptr; // column slabs of a matrix without locality (bis_spmv_slab.hip); cs_state: 0 untried, 1 in use, -1 refused
    int cs_state = 0;
    double cs_trial_ms[2] = {0.0, 0.0};        // the build-time trial: one pass / the K passes
    int sw8_state = 0;
    // second table for the SpMV with the fused (y,w) epilogue (CG): larger blocks win there
    int32_t *blkf_row = nullptr;
    int64_t *blkf_nnz = nullptr;
    int n_blocks_f = 0;
    int chunk_f = 0;
    int chunk_nnz = 0;        // nnz budget per block used to build blk_row
    int max_row_nnz = 0;
    int64_t max_block_nnz = 0;
    // structured-grid hint (generators, bis_mat_set_grid_hint): row = ((z*ny + y)*nx + x)*dof + d; 0 = none.
    // Inherited by the strict triangles and the ILU(0) factors; the tiled sweep cuts its tiles in all grid directions with it.
    int64_t grid[4] = {0, 0, 0, 0};
    // triangular-solve plans (built lazily)
    bis_trsv_plan *plan_fwd = nullptr;
    bis_trsv_plan *plan_bwd = nullptr;
    // the tiled natural-order sweep's plans (bis_trsv_tiled.hip), tried first: where they exist no level analysis is made
    struct bis_trsv_tiled *tiled_fwd = nullptr, *tiled_bwd = nullptr;
    bool tiled_tried_fwd = false, tiled_tried_bwd = false;
    // the chained sweep's plans (bis_trsv_chain.hip), for matrices without a grid: built from the level analysis
    struct bis_trsv_chain *chain_fwd = nullptr, *chain_bwd = nullptr;
    bool chain_tried_fwd = false, chain_tried_bwd = false;
    const char *sweep_kernel[2] = {"", ""}; // the kernel the last forward / backward sweep on this triangle ran (bis_mat_sweep_kernel)
};

#define BIS_HIP_CHECK(ctx, call)                                               \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) {                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);    \
            return BIS_ERR_HIP;                                                \
        }                                                                      \
    } while (0)

// after a stream synchronisation: did a kernel raise the fault word?
bis_status bis_fault_check(bis_ctx *ctx);
#define BIS_SYNC_CHECK(ctx)                                                    \
    do {                                                                       \
        BIS_HIP_CHECK(ctx, hipStreamSynchronize((ctx)->stream));               \
        if (bis_status fs_ = bis_fault_check(ctx)) return fs_;                 \
    } while (0)

#define BIS_REQUIRE(ctx, cond, msg)                                            \
    do {                                                                       \
        if (!(cond)) {                                                         \
            if (ctx) (ctx)->err = (msg);                                       \
            return BIS_ERR_INVALID;                                            \
        }                                                                      \
    } while (0)

#define BIS_CTX_OK(ctx)                                                        \
    do {                                                                       \
        if (!(ctx)) return BIS_ERR_NO_DEVICE;                                  \
    } while (0)

// ---- device helpers ---------------------------------------------------------
// wave64 sum via DPP-free shuffles (ds_bpermute under the hood for doubles).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Block sum for blockDim.x == T (multiple of 64). Result valid in thread 0.
template <int T>
__device__ __forceinline__ double block_sum(double v, double *lds /*[T/64]*/) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < T / 64; ++i) r += lds[i];
    }
    return r;
}

// XCD-aware remap (cdna_hip_programming.md T1 / MI355X_MICROARCH.md "Workgroup
// dispatch"): hardware deals workgroups round-robin over the 8 XCDs, so
// blockIdx b lands on XCD b%8.  Map it to a logical block id such that each
// XCD sweeps one contiguous range of logical blocks (its private L2 then sees
// neighbouring rows / x-planes).  Valid for any n; ids >= n are skipped by the
// caller.  Placement is a speed matter only.
__device__ __forceinline__ int xcd_remap(int b, int n_padded8) {
    const int per = n_padded8 >> 3;
    return (b & 7) * per + (b >> 3);
}
// Grouped form: within every window of 8*G consecutive logical blocks, XCD j
// takes the G consecutive blocks [j*G, (j+1)*G).  G = 1 is blockIdx order.
__device__ __forceinline__ int xcd_group_remap(int b, int G) {
    const int j = b & 7, q = b >> 3;
    return (q / G) * (8 * G) + j * G + (q % G);
}

// HIP-event bracket around one launch (bis_profile_enable)
inline void bis_prof_begin(bis_ctx *ctx) {
    if (!ctx->profile) return;
    if (ctx->prof_used == ctx->prof_events.size()) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        ctx->prof_events.emplace_back(a, b);
    }
    hipEventRecord(ctx->prof_events[ctx->prof_used].first, ctx->stream);
}
inline void bis_prof_end(bis_ctx *ctx) {
    if (!ctx->profile) return;
    hipEventRecord(ctx->prof_events[ctx->prof_used].second, ctx->stream);
    ++ctx->prof_used;
}

// result_dev[v] = sum_i partials[v*stride + i], i < n_partials (fixed order)
bis_status bis_reduce_finish(bis_ctx *ctx, int n_partials, int n_values,
                             size_t stride, double *result_dev);
// make sure ctx->partials holds at least n doubles (stream-synchronising
// only when it has to grow)
bis_status bis_ensure_partials(bis_ctx *ctx, size_t n);
// y = A x; if w != nullptr also partials[partials_off + b] = sum_{r in block b}
// y[r]*w[r] (n_partials = number written).  The caller sizes ctx->partials.
bis_status bis_spmv_launch(bis_ctx *ctx, const bis_mat *A, const double *x,
                           double *y, const double *w, int *n_partials,
                           size_t partials_off = 0);
bis_status bis_spmv_trsv_level(bis_ctx *ctx, const bis_mat *T, const double *x, double *y,
                               const double *b, const double *D);
// rows [ra,rb) of A as a matrix sharing A's arrays (y must be offset by ra)
bis_status bis_mat_row_view(bis_ctx *ctx, const bis_mat *A, int64_t ra,
                            int64_t rb, bis_mat **out);

// internal launchers shared between translation units
bis_status bis_mat_alloc(bis_ctx *ctx, int64_t n_rows, int64_t n_cols,
                         int64_t nnz, bool rp64, bis_mat **out);
bis_status bis_mat_finalize(bis_ctx *ctx, bis_mat *A);
bis_status bis_spmv_build_window(bis_ctx *ctx, bis_mat *A);
void bis_spmv_drop_packed(bis_mat *A);
void bis_spmv_drop_valdict(bis_mat *A); // after the values of A changed in place
bis_status bis_spmv_try_valdict(bis_ctx *ctx, bis_mat *A, bool consecutive_ok);
// x-window + sliced-ELL dictionary form (bis_spmv_sell.hip)
bis_status bis_spmv_sellwin_try(bis_ctx *ctx, bis_mat *A);
int bis_spmv_sellwin_blocks(const bis_mat *A);
int64_t bis_spmv_sellwin_bytes(const bis_mat *A);
int64_t bis_spmv_sellwin_slices(const bis_mat *A);
int bis_spmv_sellwin_format(const bis_mat *A);
bis_status bis_spmv_sellwin_launch(bis_ctx *ctx, const bis_mat *A, const double *x, double *y, int mode, const double *w,
                                   double *partials, const int *stop, int remap_arg, int grid);
void bis_spmv_sellwin_drop(bis_mat *A);
// window + sliced-ELL form with the 8-byte values streamed (bis_spmv_sell.hip, "win8")
bis_status bis_spmv_win8_try(bis_ctx *ctx, bis_mat *A);
int bis_spmv_win8_blocks(const bis_mat *A);
int64_t bis_spmv_win8_slices(const bis_mat *A);
int64_t bis_spmv_win8_partials(const bis_mat *A);
int64_t bis_spmv_win8_bytes(const bis_mat *A);
bis_status bis_spmv_win8_launch(bis_ctx *ctx, const bis_mat *A, const double *x, double *y, int mode, const double *w,
                                double *partials, const int *stop, int remap_arg, int grid);
void bis_spmv_win8_drop(bis_mat *A);
bis_status bis_spmv_colslab_build(bis_ctx *ctx, const bis_mat *A, int K, std::vector<bis_mat *> &slabs, bool *ok);
void bis_spmv_colslab_free(bis_ctx *ctx, std::vector<bis_mat *> &slabs);
void bis_spmv_colslab_drop(bis_mat *A);
int bis_spmv_remap_arg(int nb);
int bis_spmv_grid(int nb);
size_t bis_spmv_win8_stream_bytes(const bis_mat *A);
void *bis_spmv_win8_swap_stream(bis_mat *A, void *stream); // returns the buffer that was in use
// try to build the packed-column stream of table t (0 plain, 1 fused); A->pk_state[t] tells the outcome
bis_status bis_spmv_try_pack(bis_ctx *ctx, bis_mat *A, int t);
// free row-block tables, packed streams and window structures (not the CRS arrays)
void bis_mat_free_meta(bis_mat *A);
// after the values of A changed in place: drops every structure derived from them (dictionaries, code streams, sweep plans)
void bis_mat_values_changed(bis_mat *A);
void bis_trsv_plan_destroy(bis_trsv_plan *p);
// `to` has the sparsity pattern of `from` (ILU(0): the factor L and the strict lower triangle of A the elimination was scheduled
// by): it takes over from's level plan (levels, level-sorted rows) instead of analysing the same pattern again.  No-op when
// from has none, to has one, or the plan refers to from's storage (row views).
void bis_trsv_plan_adopt(bis_mat *to, bis_mat *from, bool backward);
bis_status bis_mat_split_strict_impl(bis_ctx *ctx, const bis_mat *A, bis_mat **L_strict,
                                     bis_mat **U_strict, double *D, double *D_inv, bool check_diag);
// tiled natural-order sweep (bis_trsv_tiled.hip); *out stays null when the matrix does not qualify
struct bis_trsv_tiled;
bis_status bis_trsv_tiled_build(bis_ctx *ctx, const bis_mat *T, bool backward, bis_trsv_tiled **out);
bis_status bis_trsv_tiled_solve(bis_ctx *ctx, bis_trsv_tiled *p, double *x, const double *D, const double *b);
void bis_trsv_tiled_destroy(bis_trsv_tiled *p);
// device-side level analysis of a strictly triangular matrix (bis_analysis.hip); level_out (optional): the level of
// every row, a device array the caller frees
bis_status bis_trsv_analyse_device(bis_ctx *ctx, const bis_mat *T, bool backward, int32_t *perm_dev,
                                   std::vector<int64_t> &level_ptr, int &n_levels, int64_t &max_width,
                                   bool &triangular, int **level_out = nullptr);
// chained sweep (bis_trsv_chain.hip); *out stays null where it does not apply
struct bis_trsv_chain;
bis_status bis_trsv_chain_build(bis_ctx *ctx, const bis_mat *T, bool backward, const int *level_dev, int n_levels,
                                bis_trsv_chain **out);
bis_status bis_trsv_chain_solve(bis_ctx *ctx, const bis_mat *T, bis_trsv_chain *p, double *x, const double *D, const double *b);
void bis_trsv_chain_destroy(bis_trsv_chain *p);
int bis_trsv_chain_resident_pairs(const bis_ctx *ctx, bool rp64, bool backward); // wave pairs (= chains in flight) of a full grid
// contiguous independent row blocks of a strictly triangular matrix, in processing order (bis_analysis.hip)
bis_status bis_trsv_blocks_device(bis_ctx *ctx, const bis_mat *T, bool backward, int max_blocks,
                                  std::vector<int64_t> &bounds, int32_t *perm_dev, bool &triangular);
// dependency levels of a strict-lower matrix: host level boundaries + device row list
bis_status bis_trsv_level_sets(bis_ctx *ctx, const bis_mat *T_lower, const std::vector<int64_t> **level_ptr,
                               const int32_t **perm_dev);
