// gauss_seidel.hpp -- Gauss-Seidel and symmetric Gauss-Seidel as solvers,
// reference methods/gauss_seidel.hpp:26-52, :76-105, :119-129.  The sweeps are
// the device triangular solves; tmp aliases as in the reference (:34, :48).
#pragma once

#include "../solver.hpp"
#include "jacobi.hpp" // run_stat_schedule

inline void gs_separate_iteration(Timers *timers, const MatrixCRS *U, const MatrixCRS *L, double *tmp,
                                  const double *D, const double *b, double *x) {
    TIME(timers, "spmv", spmv(U, x, tmp))                              // tmp <- U x
    TIME(timers, "sum", subtract_vectors(tmp, b, tmp, U->n_rows))      // tmp <- b - tmp
    TIME(timers, "sptrsv", sptrsv(L, x, D, tmp))                       // x <- (D+L)^-1 tmp
}
inline void bgs_separate_iteration(Timers *timers, const MatrixCRS *U, const MatrixCRS *L, double *tmp,
                                   const double *D, const double *b, double *x) {
    TIME(timers, "spmv", spmv(L, x, tmp))                              // tmp <- L x
    TIME(timers, "sum", subtract_vectors(tmp, b, tmp, L->n_rows))
    TIME(timers, "sptrsv", bsptrsv(U, x, D, tmp))                      // x <- (D+U)^-1 tmp
}

class GaussSeidelSolver : public Solver {
  public:
    double *x = nullptr;
    // device schedule (bis_stat_*, the default; -unfused: the reference's blocking norm per iteration): the same
    // operations in the same order, norm and stopping test on the device, sweeps and SpMVs no-ops once it has fired
    bool fused = false;
    bis_stat *fst = nullptr;
    std::vector<double> fused_hist;
    explicit GaussSeidelSolver(const Args *a) : Solver(a) { fused = !a->unfused && residual_check_len == 1; }
    virtual int stat_kind() const { return BIS_STAT_GS; }
    void allocate_structs(const int n) override {
        Solver::allocate_structs(n);
        x = dalloc(n);
    }
    void init_structs(const int n) override {
        Solver::init_structs(n);
        copy_vector(x, x_0, n);
    }
    void init_residual() override {
        if (fused) {
            bis::check(bis_stat_create(bis::ctx(), stat_kind(), A->dev, L_strict->dev, U_strict->dev, A_D, b, x, &fst), "bis_stat_create");
            bis::check(bis_stat_init(bis::ctx(), fst, tolerance, &residual_norm), "bis_stat_init");
            collected_residual_norms[collected_residual_norms_count++] = residual_norm;
            return;
        }
        compute_residual(A.get(), x, b, residual, tmp);
        residual_norm = euclidean_vec_norm(residual, N);
        Solver::init_residual();
    }
    void iterate(Timers *timers) override {
        if (fused) { // (chunks of 8: the launches behind the stopping iteration are no-ops, but many)
            if (fused_hist.empty()) run_stat_schedule(timers, fst, max_iters, 8, fused_hist);
            return;
        }
        gs_separate_iteration(timers, U_strict.get(), L_strict.get(), tmp, A_D, b, x);
    }
    void exchange() override {}
    void save_x_star() override {
        std::swap(x, x_star);
        Solver::save_x_star();
    }
    void record_residual_norm() override {
        if (fused) {
            residual_norm = fused_hist[std::min<size_t>(iter_count, fused_hist.size() - 1)];
        } else {
            compute_residual(A.get(), x, b, residual, tmp);
            residual_norm = euclidean_vec_norm(residual, N);
        }
        Solver::record_residual_norm();
    }
    ~GaussSeidelSolver() override {
        if (fst) bis_stat_destroy(bis::ctx(), fst);
        dfree(x);
    }
};

class SymmetricGaussSeidelSolver : public GaussSeidelSolver {
  public:
    explicit SymmetricGaussSeidelSolver(const Args *a) : GaussSeidelSolver(a) {}
    int stat_kind() const override { return BIS_STAT_SGS; }
    void iterate(Timers *timers) override {
        if (fused) { GaussSeidelSolver::iterate(timers); return; }
        gs_separate_iteration(timers, U_strict.get(), L_strict.get(), tmp, A_D, b, x);
        bgs_separate_iteration(timers, U_strict.get(), L_strict.get(), tmp, A_D, b, x);
    }
};
