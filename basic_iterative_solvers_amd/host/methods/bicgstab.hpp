// bicgstab.hpp -- left-preconditioned BiCGSTAB, reference methods/bicgstab.hpp:
// 8-83 (iteration), :147-169 (init_residual: r_0-tilde and p_0 are the
// PRECONDITIONED initial residual, rho_0 = (r_0, M^-1 r_0)), :171-223.
#pragma once

#include "../solver.hpp"

inline void bicgstab_separate_iteration(Timers *timers, const PrecondType pc, const MatrixCRS *A,
                                        const MatrixCRS *L, const MatrixCRS *U, double *A_D,
                                        double *A_D_inv, double *L_D, double *U_D, double *x_new,
                                        double *x_old, double *tmp, double *work, double *p_new,
                                        double *p_old, double *r_new, double *r_old, double *r_0,
                                        double *v, double *h, double *s, double *s_tmp, double *y,
                                        double *z, double &rho_new, double rho_old) {
    const int N = A->n_cols;
    TIME(timers, "precond", apply_preconditioner(pc, N, L, U, A_D, A_D_inv, L_D, U_D, y, p_old, tmp, work))
    TIME(timers, "spmv", spmv(A, y, v))
    double alpha;
    TIME(timers, "dot", alpha = rho_old / dot(r_0, v, N))
    TIME(timers, "sum", subtract_vectors(s, r_old, v, N, alpha))
    TIME(timers, "precond", apply_preconditioner(pc, N, L, U, A_D, A_D_inv, L_D, U_D, s_tmp, s, tmp, work))
    TIME(timers, "spmv", spmv(A, s_tmp, z))
    double omega;
    TIME(timers, "dot", omega = dot(z, s, N) / dot(z, z, N))
    TIME(timers, "sum", sum_vectors(h, x_old, y, N, alpha))
    TIME(timers, "sum", sum_vectors(x_new, h, s_tmp, N, omega))
    TIME(timers, "sum", subtract_vectors(r_new, s, z, N, omega))
    TIME(timers, "dot", rho_new = dot(r_0, r_new, N))
    const double beta = (rho_new / rho_old) * (alpha / omega);
    TIME(timers, "sum", subtract_vectors(tmp, p_old, v, N, omega))
    TIME(timers, "sum", sum_vectors(p_new, r_new, tmp, N, beta))
}

class BiCGSTABSolver : public Solver {
  public:
    double *x_new = nullptr, *x_old = nullptr, *p_old = nullptr, *p_new = nullptr, *v = nullptr,
           *h = nullptr, *s = nullptr, *s_tmp = nullptr, *y = nullptr, *z = nullptr,
           *residual_old = nullptr, *residual_new = nullptr;
    double rho_old = 0.0, rho_new = 0.0;
    explicit BiCGSTABSolver(const Args *a) : Solver(a) {}
    void allocate_structs(const int n) override {
        Solver::allocate_structs(n);
        double **vv[] = {&x_new, &x_old, &p_new, &p_old, &residual_new, &residual_old, &v, &h, &s, &s_tmp, &y, &z};
        for (auto p : vv) *p = dalloc(n);
    }
    void init_structs(const int n) override {
        Solver::init_structs(n);
        double *zz[] = {x_new, p_new, p_old, residual_new, residual_old, v, h, s, s_tmp, y, z};
        for (auto p : zz) init_vector(p, 0.0, n);
        copy_vector(x_old, x_0, n);
    }
    void init_residual() override {
        compute_residual(A.get(), x_old, b, residual, tmp);
        copy_vector(residual_old, residual, N);
        residual_norm = euclidean_vec_norm(residual, N);
        apply_preconditioner(preconditioner, N, L_strict.get(), U_strict.get(), A_D, A_D_inv, L_D, U_D,
                             residual, residual, tmp, work); // in place
        copy_vector(p_old, residual, N);
        rho_old = dot(residual_old, residual, N);
        Solver::init_residual();
    }
    void iterate(Timers *timers) override {
        bicgstab_separate_iteration(timers, preconditioner, A.get(), L_strict.get(), U_strict.get(), A_D,
                                    A_D_inv, L_D, U_D, x_new, x_old, tmp, work, p_new, p_old, residual_new,
                                    residual_old, residual_0, v, h, s, s_tmp, y, z, rho_new, rho_old);
        std::swap(residual, residual_new);
    }
    void exchange() override {
        std::swap(p_old, p_new);
        std::swap(residual_old, residual);
        std::swap(x_old, x_new);
        std::swap(rho_old, rho_new);
    }
    void save_x_star() override {
        std::swap(x_old, x_star);
        Solver::save_x_star();
    }
    void record_residual_norm() override {
        residual_norm = euclidean_vec_norm(residual, N);
        Solver::record_residual_norm();
    }
    ~BiCGSTABSolver() override {
        double *vv[] = {x_new, x_old, p_new, p_old, residual_new, residual_old, v, h, s, s_tmp, y, z};
        for (auto p : vv) dfree(p);
    }
};
