// gauss_seidel.hpp -- Gauss-Seidel and symmetric Gauss-Seidel as solvers,
// reference methods/gauss_seidel.hpp:26-52, :76-105, :119-129.  The sweeps are
// the device triangular solves; tmp aliases as in the reference (:34, :48).
#pragma once

#include "../solver.hpp"

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
    explicit GaussSeidelSolver(const Args *a) : Solver(a) {}
    void allocate_structs(const int n) override {
        Solver::allocate_structs(n);
        x = dalloc(n);
    }
    void init_structs(const int n) override {
        Solver::init_structs(n);
        copy_vector(x, x_0, n);
    }
    void init_residual() override {
        compute_residual(A.get(), x, b, residual, tmp);
        residual_norm = euclidean_vec_norm(residual, N);
        Solver::init_residual();
    }
    void iterate(Timers *timers) override {
        gs_separate_iteration(timers, U_strict.get(), L_strict.get(), tmp, A_D, b, x);
    }
    void exchange() override {}
    void save_x_star() override {
        std::swap(x, x_star);
        Solver::save_x_star();
    }
    void record_residual_norm() override {
        compute_residual(A.get(), x, b, residual, tmp);
        residual_norm = euclidean_vec_norm(residual, N);
        Solver::record_residual_norm();
    }
    ~GaussSeidelSolver() override { dfree(x); }
};

class SymmetricGaussSeidelSolver : public GaussSeidelSolver {
  public:
    explicit SymmetricGaussSeidelSolver(const Args *a) : GaussSeidelSolver(a) {}
    void iterate(Timers *timers) override {
        gs_separate_iteration(timers, U_strict.get(), L_strict.get(), tmp, A_D, b, x);
        bgs_separate_iteration(timers, U_strict.get(), L_strict.get(), tmp, A_D, b, x);
    }
};
