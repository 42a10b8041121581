// common.hpp -- host-side common definitions of the MI355X build: the enums,
// CLI argument block, compile-time configuration and timer tree that the
// reference keeps in common.hpp (:38-111, :206-354), plus the glue to the
// C-ABI device library (include/bis_hip.h).  Vectors handled by this layer are
// DEVICE pointers obtained from dalloc(); nothing here computes on the CPU.
#pragma once

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <sys/time.h>
#include <vector>

#include "bis_hip.h"

// Compile-time configuration: same names and defaults as the reference's
// CMake cache (CMakeLists.txt:19-29).
#ifndef MAX_ITERS
#define MAX_ITERS 1000
#endif
#ifndef TOL
#define TOL 1e-14
#endif
#ifndef RES_CHECK_LEN
#define RES_CHECK_LEN 1
#endif
#ifndef PRECOND_OUTER_ITERS
#define PRECOND_OUTER_ITERS 1
#endif
#ifndef PRECOND_INNER_ITERS
#define PRECOND_INNER_ITERS 0
#endif
#ifndef INIT_X_VAL
#define INIT_X_VAL 0.1
#endif
#ifndef B_VAL
#define B_VAL 1.0
#endif
#ifndef ILU0_PIVOT_TOLERANCE
#define ILU0_PIVOT_TOLERANCE 1e-8
#endif
#ifndef ILU0_PIVOT_REPLACEMENT
#define ILU0_PIVOT_REPLACEMENT 1e-4
#endif

using Interface = void *; // the SMAX seam of the reference (common.hpp:18-24) is the C ABI here
#define SMAX_ARGS(...)

enum class PrecondType { // ordinals == BIS_PC_* == reference common.hpp:38-47
    None, Jacobi, GaussSeidel, BackwardsGaussSeidel, SymmetricGaussSeidel, TwoStageGS,
    SymmetricTwoStageGS, ILU0
};
enum class SolverType { Jacobi, GaussSeidel, SymmetricGaussSeidel, GMRES, ConjugateGradient, BiCGSTAB };

inline std::string to_string(PrecondType t) {
    static const char *n[] = {"none", "jacobi", "gauss-seidel", "backwards-gauss-seidel",
                              "symmetric-gauss-seidel", "two-stage gauss-seidel",
                              "symmetric two-stage gauss-seidel", "incomplete LU(0)"};
    return n[static_cast<int>(t)];
}
inline std::string to_string(SolverType t) {
    static const char *n[] = {"jacobi", "gauss-seidel", "symmetric-gauss-seidel", "gmres",
                              "conjugate-gradient", "bicgstab"};
    return n[static_cast<int>(t)];
}

struct Args {
    std::string matrix_file_name{};
    SolverType method{};
    PrecondType preconditioner{};
    int restart_length = 10;
    bool num_scale = false;
    bool unfused = false; // -unfused: CG / Jacobi / GS / SGS run the reference's kernel-by-kernel schedule (blocking reductions) instead of the device schedules
    bool host_scalars = false; // -hostscalars: GMRES / BiCGSTAB return every dot product to the host like the reference
                               // (default: Gram-Schmidt coefficients, alpha / omega / beta stay on the device)
    std::string perm_mode = "none"; // -perm mc: multi-colour reordering (SMAX PERM_MODE role)
    std::string dump_perm;          // -dump-perm FILE: write perm[new]=old
    std::string dump_x;             // -dump-x FILE: write x* (natural row order, also after -perm)
    std::string crs_cache;          // -cache FILE: binary CRS next to a .mtx input (read if present, else written)
    bool perm_host = false;         // -perm-host: colour and permute on the host (fallback path)
    int trsv_mode = -1;             // -trsv tiled|level|chain|wave: natural-order sweeps with the tiled kernel also where its plan is built on the host / never (default: tiled where the matrix has a grid hint)
    long long grid_hint[4] = {0, 0, 0, 0}; // -grid NX,NY,NZ[,DOF]: a matrix read from a file is a stencil on this grid, x fastest (bis_mat_set_grid_hint)
    int device = 0;
};

// ---- device glue ---------------------------------------------------------------
namespace bis {
inline bis_ctx *&ctx_slot() { static bis_ctx *c = nullptr; return c; }
inline bis_ctx *ctx() {
    if (!ctx_slot()) {
        fprintf(stderr, "ERROR: no device context (bis::init not called)\n");
        exit(EXIT_FAILURE);
    }
    return ctx_slot();
}
// the reference's kernels are void and exit on fatal conditions
// (common.hpp:382-396); restore that convention on top of bis_status
inline void check(bis_status st, const char *what) {
    if (st != BIS_OK) {
        fprintf(stderr, "ERROR: %s: %s (status %d)\n", what, bis_last_error(ctx_slot()), st);
        exit(EXIT_FAILURE);
    }
}
inline void init(int device) {
    bis_status st = bis_ctx_create(device, nullptr, &ctx_slot());
    if (st != BIS_OK) {
        fprintf(stderr, "ERROR: no usable gfx950 device (status %d); this build has no CPU path\n", st);
        exit(EXIT_FAILURE);
    }
}
inline void shutdown() { if (ctx_slot()) { bis_ctx_destroy(ctx_slot()); ctx_slot() = nullptr; } }
inline bool timers_sync() { static int v = getenv("BIS_TIMERS_SYNC") ? atoi(getenv("BIS_TIMERS_SYNC")) : 1; return v != 0; }
} // namespace bis

inline double *dalloc(long n) { double *p = nullptr; bis::check(bis_vec_alloc(bis::ctx(), n, &p), "bis_vec_alloc"); return p; }
inline void dfree(double *p) { if (p) bis_vec_free(bis::ctx(), p); }
inline void to_device(double *dst, const double *src, long n) { bis::check(bis_vec_upload(bis::ctx(), dst, src, n), "bis_vec_upload"); }
inline void to_host(double *dst, const double *src, long n) { bis::check(bis_vec_download(bis::ctx(), dst, src, n), "bis_vec_download"); }

// ---- timers (reference common.hpp:206-354, utilities.hpp:110-152) -----------------
class Stopwatch {
    long double wtime = 0;
    timeval begin{}, end{};
  public:
    void start() { gettimeofday(&begin, 0); }
    void stop() {
        // kernels are asynchronous: drain the stream so the timer tree means
        // what it means in the reference (BIS_TIMERS_SYNC=0 turns this off)
        if (bis::timers_sync() && bis::ctx_slot()) bis_sync(bis::ctx_slot());
        gettimeofday(&end, 0);
        wtime += (end.tv_sec - begin.tv_sec) + (end.tv_usec - begin.tv_usec) * 1e-6;
    }
    long double check() {
        gettimeofday(&end, 0);
        return (end.tv_sec - begin.tv_sec) + (end.tv_usec - begin.tv_usec) * 1e-6;
    }
    long double get_wtime() const { return wtime; }
};

struct Timers {
    std::map<std::string, Stopwatch> t;
    Stopwatch &operator[](const std::string &k) { return t[k]; }
    Stopwatch *per_iteration_time = &t["per_iteration"];
};

#define TIME(timers, name, routine)                                                                \
    do {                                                                                           \
        Stopwatch &sw_ = (*(timers))[name];                                                        \
        sw_.start();                                                                               \
        routine;                                                                                   \
        sw_.stop();                                                                                \
    } while (0);

#ifdef DEBUG_MODE
#define IF_DEBUG_MODE(s) s;
#else
#define IF_DEBUG_MODE(s)
#endif

struct SanityChecker { // reference common.hpp:388-396
    static void zero_diag(int row) { fprintf(stderr, "Zero detected on diagonal at row index %d\n", row); exit(EXIT_FAILURE); }
    static void no_diag(int row) { fprintf(stderr, "No diagonal to extract at row index %d\n", row); exit(EXIT_FAILURE); }
};
