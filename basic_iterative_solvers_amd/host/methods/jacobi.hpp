// jacobi.hpp -- Jacobi iteration, reference methods/jacobi.hpp:43-52 (iteration),
// :79-107 (residual bookkeeping: a full true residual every iteration).
#pragma once

#include "../solver.hpp"

inline void jacobi_separate_iteration(Timers *timers, const MatrixCRS *A, const double *D,
                                      const double *b, double *x_new, const double *x_old) {
    TIME(timers, "spmv", spmv(A, x_old, x_new))                         // x_new <- A x_old
    TIME(timers, "normalize", normalize_x(x_new, x_old, D, b, A->n_rows)) // x_new <- (b-(x_new-D x_old))/D
}

class JacobiSolver : public Solver {
  public:
    double *x_new = nullptr, *x_old = nullptr;
    explicit JacobiSolver(const Args *a) : Solver(a) {}
    void allocate_structs(const int n) override {
        Solver::allocate_structs(n);
        x_new = dalloc(n);
        x_old = dalloc(n);
    }
    void init_structs(const int n) override {
        Solver::init_structs(n);
        init_vector(x_new, 0.0, n);
        copy_vector(x_old, x_0, n);
    }
    void init_residual() override {
        compute_residual(A.get(), x_old, b, residual, tmp);
        residual_norm = euclidean_vec_norm(residual, N);
        Solver::init_residual();
    }
    void iterate(Timers *timers) override { jacobi_separate_iteration(timers, A.get(), A_D, b, x_new, x_old); }
    void exchange() override { std::swap(x_old, x_new); }
    void save_x_star() override {
        std::swap(x_old, x_star);
        Solver::save_x_star();
    }
    void record_residual_norm() override {
        compute_residual(A.get(), x_new, b, residual, tmp);
        residual_norm = euclidean_vec_norm(residual, N);
        Solver::record_residual_norm();
    }
    ~JacobiSolver() override { dfree(x_new); dfree(x_old); }
};
