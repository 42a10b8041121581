// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref build).
//
// A thin extern "C" shim around the REAL reference (DanecLacey/
// basic_iterative_solvers), compiled from the reference's own headers where
// they lie under $(REF) (= /root/reference) by oracle/Makefile into
// oracle/_ref/libbisref*.so.  No reference source is copied: this file only
// #includes the reference headers and forwards to the reference's functions.
// It exists so that (a) the C restatement in oracle/bis_oracle.c can be
// validated against the reference itself and (b) golden vectors under
// tests/golden/ can be generated from the reference (tests/golden/
// make_golden.py).  It is never linked into the product.
//
// Compile-time configuration macros are the reference's CMake defaults
// (CMakeLists.txt:19-29, :232-243); see oracle/Makefile.
#include <unistd.h>

#include "common.hpp"
#include "kernels.hpp"
#include "methods/bicgstab.hpp"
#include "methods/cg.hpp"
#include "methods/gauss_seidel.hpp"
#include "methods/gmres.hpp"
#include "methods/jacobi.hpp"
#include "postprocessing.hpp"
#include "preprocessing.hpp"
#include "solver_harness.hpp"
#include "sparse_matrix.hpp"
#include "utilities/utilities.hpp"

#include <cstdint>
#include <cstring>
#include <memory>

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {

// Borrow caller-owned arrays as a MatrixCRS for the duration of a call.
struct Borrowed {
    MatrixCRS m;
    Borrowed(int n_rows, int n_cols, int nnz, const int *rp, const int *col,
             const double *val) {
        m.n_rows = n_rows;
        m.n_cols = n_cols;
        m.nnz = nnz;
        m.row_ptr = const_cast<int *>(rp);
        m.col = const_cast<int *>(col);
        m.val = const_cast<double *>(val);
    }
    ~Borrowed() { m.row_ptr = nullptr; m.col = nullptr; m.val = nullptr; }
};

// 0: the matrix copy is written by the calling thread (memcpy: what a caller
// that hands the reference host arrays gets).  1: through the reference's own
// MatrixCRS::operator= (sparse_matrix.hpp:92-128), whose copy loop is an
// `omp parallel for schedule(static)` over the rows -- the rows are first
// touched by the threads that later multiply them (bench.py's
// `cpu_baseline_first_touch` leg).
int g_first_touch = 0;

std::unique_ptr<MatrixCRS> own_copy(int n, int nnz, const int *rp,
                                    const int *col, const double *val) {
    if (g_first_touch) {
        Borrowed src(n, n, nnz, rp, col, val);
        auto A = std::make_unique<MatrixCRS>();
        *A = src.m;
        return A;
    }
    auto A = std::make_unique<MatrixCRS>(n, n, nnz);
    std::memcpy(A->row_ptr, rp, sizeof(int) * (n + 1));
    std::memcpy(A->col, col, sizeof(int) * nnz);
    std::memcpy(A->val, val, sizeof(double) * nnz);
    return A;
}

// The harness prints milestone lines + timer trees to stdout; keep the host
// process' stdout clean while the reference runs.
struct QuietStdout {
    int saved;
    QuietStdout() {
        fflush(stdout);
        std::cout.flush();
        saved = dup(1);
        int nul = open_null();
        dup2(nul, 1);
        close(nul);
    }
    static int open_null();
    ~QuietStdout() {
        fflush(stdout);
        std::cout.flush();
        dup2(saved, 1);
        close(saved);
    }
};
} // namespace
#include <fcntl.h>
int QuietStdout::open_null() { return open("/dev/null", O_WRONLY); }

REF_API void ref_set_first_touch(int on) { g_first_touch = on; }

REF_API int ref_config(int *out6, double *outd5) {
    out6[0] = MAX_ITERS;
    out6[1] = RES_CHECK_LEN;
    out6[2] = PRECOND_OUTER_ITERS;
    out6[3] = PRECOND_INNER_ITERS;
    out6[4] = 0;
    out6[5] = 0;
    outd5[0] = TOL;
    outd5[1] = INIT_X_VAL;
    outd5[2] = B_VAL;
    outd5[3] = ILU0_PIVOT_TOLERANCE;
    outd5[4] = ILU0_PIVOT_REPLACEMENT;
    return 0;
}

// ---- kernels.hpp -----------------------------------------------------------
REF_API void ref_spmv(int n_rows, int nnz, const int *rp, const int *col,
                      const double *val, const double *x, double *y) {
    Borrowed A(n_rows, n_rows, nnz, rp, col, val);
    native_spmv(&A.m, x, y);
}
REF_API void ref_sptrsv(int n_rows, int nnz, const int *rp, const int *col,
                        const double *val, double *x, const double *D,
                        const double *b) {
    Borrowed L(n_rows, n_rows, nnz, rp, col, val);
    native_sptrsv(&L.m, x, D, b);
}
REF_API void ref_bsptrsv(int n_rows, int nnz, const int *rp, const int *col,
                         const double *val, double *x, const double *D,
                         const double *b) {
    Borrowed U(n_rows, n_rows, nnz, rp, col, val);
    native_bsptrsv(&U.m, x, D, b);
}
REF_API void ref_subtract_vectors(double *r, const double *a, const double *b,
                                  int n, double s) {
    subtract_vectors(r, a, b, n, s);
}
REF_API void ref_sum_vectors(double *r, const double *a, const double *b,
                             int n, double s) {
    sum_vectors(r, a, b, n, s);
}
REF_API void ref_elemwise_mult_vectors(double *r, const double *a,
                                       const double *b, int n, double s) {
    elemwise_mult_vectors(r, a, b, n, s);
}
REF_API void ref_elemwise_div_vectors(double *r, const double *a,
                                      const double *b, int n, double s) {
    elemwise_div_vectors(r, a, b, n, s);
}
REF_API double ref_dot(const double *a, const double *b, int n) {
    return dot(a, b, n);
}
REF_API double ref_euclidean_vec_norm(const double *v, int n) {
    return euclidean_vec_norm(v, n);
}
REF_API void ref_scale(double *r, const double *v, double s, int n) {
    scale(r, v, s, n);
}
REF_API void ref_copy_vector(double *o, const double *i, int n) {
    copy_vector(o, i, n);
}
REF_API void ref_init_vector(double *v, double val, long n) {
    init_vector(v, val, n);
}
REF_API void ref_normalize_x(double *x_new, const double *x_old,
                             const double *D, const double *b, int n) {
    normalize_x(x_new, x_old, D, b, n);
}
// dgemm_transpose1 exactly as gmres.hpp:358 calls it (n_vec columns).
REF_API void ref_dgemm_transpose1(double *V, double *y, double *Vy, int N,
                                  int n_vec) {
    dgemm_transpose1(V, y, Vy, N, n_vec, 1);
}
REF_API void ref_compute_residual(int n_rows, int nnz, const int *rp,
                                  const int *col, const double *val,
                                  const double *x, const double *b,
                                  double *res, double *tmp) {
    Borrowed A(n_rows, n_rows, nnz, rp, col, val);
    compute_residual(&A.m, x, b, res, tmp);
}
REF_API void ref_apply_preconditioner(
    int pc, int N, int nnzL, const int *Lrp, const int *Lcol,
    const double *Lval, int nnzU, const int *Urp, const int *Ucol,
    const double *Uval, double *A_D, double *A_D_inv, double *L_D, double *U_D,
    double *out, double *in, double *tmp, double *work) {
    Borrowed L(N, N, nnzL, Lrp, Lcol, Lval);
    Borrowed U(N, N, nnzU, Urp, Ucol, Uval);
    apply_preconditioner(static_cast<PrecondType>(pc), N, &L.m, &U.m, A_D,
                         A_D_inv, L_D, U_D, out, in, tmp, work);
}

// ---- setup -----------------------------------------------------------------
// read_from_mtx + convert_coo_to_crs.  Two-call protocol: sizes, then fill.
static std::unique_ptr<MatrixCRS> g_last;
REF_API int ref_read_mtx(const char *path, int *n_rows, int *nnz) {
    try {
        MatrixCOO coo;
        coo.read_from_mtx(path);
        g_last = std::make_unique<MatrixCRS>();
        convert_coo_to_crs(&coo, g_last.get());
        *n_rows = g_last->n_rows;
        *nnz = g_last->nnz;
        return 0;
    } catch (const std::exception &e) {
        return -1;
    }
}
REF_API void ref_read_mtx_fetch(int *rp, int *col, double *val) {
    std::memcpy(rp, g_last->row_ptr, sizeof(int) * (g_last->n_rows + 1));
    std::memcpy(col, g_last->col, sizeof(int) * g_last->nnz);
    std::memcpy(val, g_last->val, sizeof(double) * g_last->nnz);
    g_last.reset();
}

REF_API int ref_coo_to_crs(int n_rows, int nnz, const int *I, const int *J,
                           const double *V, int *rp, int *col, double *val) {
    MatrixCOO coo(n_rows, n_rows, nnz);
    coo.I.assign(I, I + nnz);
    coo.J.assign(J, J + nnz);
    coo.values.assign(V, V + nnz);
    MatrixCRS A;
    convert_coo_to_crs(&coo, &A);
    std::memcpy(rp, A.row_ptr, sizeof(int) * (n_rows + 1));
    std::memcpy(col, A.col, sizeof(int) * nnz);
    std::memcpy(val, A.val, sizeof(double) * nnz);
    return 0;
}

// split_LU; which: 0 L, 1 L_strict, 2 U, 3 U_strict.  Counts then fetch.
static MatrixCRS *g_split[4] = {nullptr, nullptr, nullptr, nullptr};
REF_API void ref_split_LU(int n, int nnz, const int *rp, const int *col,
                          const double *val, int *nnz4) {
    Borrowed A(n, n, nnz, rp, col, val);
    for (auto &p : g_split) { delete p; p = new MatrixCRS(); }
    split_LU(&A.m, g_split[0], g_split[1], g_split[2], g_split[3]);
    for (int k = 0; k < 4; ++k) nnz4[k] = g_split[k]->nnz;
}
REF_API void ref_split_LU_fetch(int which, int *rp, int *col, double *val) {
    MatrixCRS *M = g_split[which];
    std::memcpy(rp, M->row_ptr, sizeof(int) * (M->n_rows + 1));
    std::memcpy(col, M->col, sizeof(int) * M->nnz);
    std::memcpy(val, M->val, sizeof(double) * M->nnz);
}
// peel_diag_crs in place on caller arrays.
REF_API void ref_peel_diag_crs(int n, int nnz, const int *rp, int *col,
                               double *val, double *D, double *D_inv) {
    Borrowed A(n, n, nnz, rp, col, val);
    peel_diag_crs(&A.m, D, D_inv);
}
REF_API void ref_extract_scale(int n, int nnz, const int *rp, const int *col,
                               const double *val, double *s) {
    Borrowed A(n, n, nnz, rp, col, val);
    extract_scale(&A.m, s);
}
REF_API void ref_scale_mat(int n, int nnz, const int *rp, const int *col,
                           double *val, const double *s) {
    Borrowed A(n, n, nnz, rp, col, val);
    scale_mat(&A.m, s);
}

// The serial ILU(0) (factor_ILU0_old, LU_factors.hpp:320-539) followed by
// peel_diag_crs(U, U_D) as factor_LU does (:917-918).  Outputs L_strict,
// U_strict (sized like A's strict parts), L_D, U_D.
REF_API void ref_factor_ilu0(int n, int nnz, const int *rp, const int *col,
                             const double *val, int *Ls_rp, int *Ls_col,
                             double *Ls_val, double *L_D, int *Us_rp,
                             int *Us_col, double *Us_val, double *U_D,
                             int *nnz2) {
    Borrowed A(n, n, nnz, rp, col, val);
    Timers *timers = new Timers;
    init_timers(timers);
    MatrixCRS L, Ls, U, Us;
    factor_ILU0_old(timers, &A.m, &L, &Ls, L_D, &U, &Us, U_D);
    peel_diag_crs(&U, U_D);
    std::memcpy(Ls_rp, Ls.row_ptr, sizeof(int) * (n + 1));
    std::memcpy(Ls_col, Ls.col, sizeof(int) * Ls.nnz);
    std::memcpy(Ls_val, Ls.val, sizeof(double) * Ls.nnz);
    std::memcpy(Us_rp, Us.row_ptr, sizeof(int) * (n + 1));
    std::memcpy(Us_col, Us.col, sizeof(int) * Us.nnz);
    std::memcpy(Us_val, Us.val, sizeof(double) * Us.nnz);
    nnz2[0] = Ls.nnz;
    nnz2[1] = Us.nnz;
}

// ---- full solve --------------------------------------------------------------
// preprocessing -> solve -> (summary bookkeeping) through the reference's own
// classes.  ilu_real != 0 swaps in factor_ILU0_old for the SMAX-only
// factor_ILU0_new (SURVEY.md section 5 defect 2 / section 8c).
// out_i: [iters (as printed), n_hist, converged]; out_d: [stopping, final
// true residual, iterate seconds, sample seconds, spmv seconds] (the last three
// from the reference's Timers).  hist must hold 2*MAX_ITERS doubles, x_star n doubles.
REF_API int ref_solve(int n, int nnz, const int *rp, const int *col,
                      const double *val, int solver_type, int precond,
                      int restart_len, int num_scale, int max_iters,
                      double tol, int ilu_real, double *hist, double *x_star,
                      int *out_i, double *out_d) {
    QuietStdout quiet;
    Args args;
    args.method = static_cast<SolverType>(solver_type);
    args.preconditioner = static_cast<PrecondType>(precond);
    args.restart_length = restart_len;
    args.num_scale = num_scale != 0;
    Timers *timers = new Timers;
    init_timers(timers);
    Solver *solver = nullptr;
    switch (args.method) { // main.cpp:22-44
    case SolverType::Jacobi: solver = new JacobiSolver(&args); break;
    case SolverType::GaussSeidel: solver = new GaussSeidelSolver(&args); break;
    case SolverType::SymmetricGaussSeidel:
        solver = new SymmetricGaussSeidelSolver(&args);
        break;
    case SolverType::ConjugateGradient:
        solver = new ConjugateGradientSolver(&args);
        break;
    case SolverType::GMRES: solver = new GMRESSolver(&args); break;
    case SolverType::BiCGSTAB: solver = new BiCGSTABSolver(&args); break;
    default: return 4;
    }
    if (max_iters > 0 && max_iters <= MAX_ITERS) solver->max_iters = max_iters;
    if (tol > 0) solver->tolerance = tol;
    std::unique_ptr<MatrixCRS> A = own_copy(n, nnz, rp, col, val);

    if (!(ilu_real && args.preconditioner == PrecondType::ILU0)) {
        preprocessing(&args, solver, timers, A);
    } else {
        // preprocessing.hpp:26-100 step by step, with the serial ILU(0).
        solver->allocate_structs(A->n_cols);
        solver->init_structs(A->n_cols);
        solver->A = std::move(A);
        if (solver->num_scale) {
            int N = solver->A->n_rows;
            extract_scale(solver->A.get(), solver->A_D_scale);
            scale_mat(solver->A.get(), solver->A_D_scale);
            scale_vec(solver->x_0, solver->A_D_scale, N);
            scale_vec(solver->b, solver->A_D_scale, N);
        }
        solver->L = std::make_unique<MatrixCRS>();
        solver->L_strict = std::make_unique<MatrixCRS>();
        solver->U = std::make_unique<MatrixCRS>();
        solver->U_strict = std::make_unique<MatrixCRS>();
        split_LU(solver->A.get(), solver->L.get(), solver->L_strict.get(),
                 solver->U.get(), solver->U_strict.get());
        peel_diag_crs(solver->L.get(), solver->A_D, solver->A_D_inv);
        peel_diag_crs(solver->U.get(), solver->A_D, solver->A_D_inv);
        // fresh targets: factor_ILU0_old overwrites the array pointers
        solver->L = std::make_unique<MatrixCRS>();
        solver->L_strict = std::make_unique<MatrixCRS>();
        solver->U = std::make_unique<MatrixCRS>();
        solver->U_strict = std::make_unique<MatrixCRS>();
        factor_ILU0_old(timers, solver->A.get(), solver->L.get(),
                        solver->L_strict.get(), solver->L_D, solver->U.get(),
                        solver->U_strict.get(), solver->U_D);
        peel_diag_crs(solver->U.get(), solver->U_D);
        solver->init_residual();
        solver->init_stopping_criteria();
    }
    solve(&args, solver, timers);

    int count = solver->collected_residual_norms_count;
    for (int i = 0; i < count; ++i) hist[i] = solver->collected_residual_norms[i];
    int iters = solver->iter_count; // postprocessing.hpp:39-40
    if (solver->method == SolverType::GMRES) iters += solver->gmres_restart_count;
    out_i[0] = iters;
    out_i[1] = count;
    out_i[2] = solver->convergence_flag ? 1 : 0;
    out_d[0] = solver->stopping_criteria;
    out_d[1] = solver->collected_residual_norms[count + 1]; // solver.hpp:158
    out_d[2] = (double)timers->iterate_time->get_wtime();   // the reference's own timer tree
    out_d[3] = (double)timers->sample_time->get_wtime();
    out_d[4] = (double)timers->spmv_time->get_wtime();
    std::memcpy(x_star, solver->x_star, sizeof(double) * n);
    delete solver;
    return 0;
}
