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

// The same iteration with rho, alpha, omega, beta kept on the device (sc[]: see the enum): no host round trip inside
// the iteration; the residual norm sampled after it (record_residual_norm) is the one blocking read per iteration.
enum { BI_RHO_OLD = 0, BI_RHO_NEW, BI_D1, BI_ALPHA, BI_D2, BI_D3, BI_OMEGA, BI_BETA, BI_COUNT };
inline void bicgstab_device_scalar_iteration(Timers *timers, const PrecondType pc, const MatrixCRS *A,
                                             const MatrixCRS *L, const MatrixCRS *U, double *A_D,
                                             double *A_D_inv, double *L_D, double *U_D, double *x_new,
                                             double *x_old, double *tmp, double *work, double *p_new,
                                             double *p_old, double *r_new, double *r_old, double *r_0,
                                             double *v, double *h, double *s, double *s_tmp, double *y,
                                             double *z, double *rho_new, double *rho_old, double *sc) {
    const int N = A->n_cols;
    TIME(timers, "precond", apply_preconditioner(pc, N, L, U, A_D, A_D_inv, L_D, U_D, y, p_old, tmp, work))
    TIME(timers, "spmv", spmv(A, y, v))
    TIME(timers, "dot", { dot_dev(r_0, v, N, sc + BI_D1); scalar_div(sc + BI_ALPHA, rho_old, sc + BI_D1); })      // :34
    TIME(timers, "sum", subtract_vectors_dev(s, r_old, v, N, sc + BI_ALPHA))                                      // :39
    TIME(timers, "precond", apply_preconditioner(pc, N, L, U, A_D, A_D_inv, L_D, U_D, s_tmp, s, tmp, work))
    TIME(timers, "spmv", spmv(A, s_tmp, z))
    TIME(timers, "dot", { dot_dev(z, s, N, sc + BI_D2); dot_dev(z, z, N, sc + BI_D3);
                          scalar_div(sc + BI_OMEGA, sc + BI_D2, sc + BI_D3); })                                   // :51
    TIME(timers, "sum", sum_vectors_dev(h, x_old, y, N, sc + BI_ALPHA))                                           // :54
    TIME(timers, "sum", sum_vectors_dev(x_new, h, s_tmp, N, sc + BI_OMEGA))                                       // :61
    TIME(timers, "sum", subtract_vectors_dev(r_new, s, z, N, sc + BI_OMEGA))                                      // :64
    TIME(timers, "dot", { dot_dev(r_0, r_new, N, rho_new);
                          scalar_ratio_product(sc + BI_BETA, rho_new, rho_old, sc + BI_ALPHA, sc + BI_OMEGA); })  // :68-71
    TIME(timers, "sum", subtract_vectors_dev(tmp, p_old, v, N, sc + BI_OMEGA))                                    // :75
    TIME(timers, "sum", sum_vectors_dev(p_new, r_new, tmp, N, sc + BI_BETA))                                      // :78
}

class BiCGSTABSolver : public Solver {
  public:
    double *x_new = nullptr, *x_old = nullptr, *p_old = nullptr, *p_new = nullptr, *v = nullptr,
           *h = nullptr, *s = nullptr, *s_tmp = nullptr, *y = nullptr, *z = nullptr,
           *residual_old = nullptr, *residual_new = nullptr;
    double rho_old = 0.0, rho_new = 0.0;
    bool dev_scalars = true;
    double *sc = nullptr;                       // device scalars (BI_*)
    double *rho_old_dev = nullptr, *rho_new_dev = nullptr;
    explicit BiCGSTABSolver(const Args *a) : Solver(a), dev_scalars(!a->host_scalars) {}
    void allocate_structs(const int n) override {
        Solver::allocate_structs(n);
        double **vv[] = {&x_new, &x_old, &p_new, &p_old, &residual_new, &residual_old, &v, &h, &s, &s_tmp, &y, &z};
        for (auto p : vv) *p = dalloc(n);
        sc = dalloc(BI_COUNT);
        rho_old_dev = sc + BI_RHO_OLD;
        rho_new_dev = sc + BI_RHO_NEW;
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
        if (dev_scalars) dot_dev(residual_old, residual, N, rho_old_dev);
        else rho_old = dot(residual_old, residual, N);
        Solver::init_residual();
    }
    void iterate(Timers *timers) override {
        if (dev_scalars)
            bicgstab_device_scalar_iteration(timers, preconditioner, A.get(), L_strict.get(), U_strict.get(), A_D,
                                             A_D_inv, L_D, U_D, x_new, x_old, tmp, work, p_new, p_old, residual_new,
                                             residual_old, residual_0, v, h, s, s_tmp, y, z, rho_new_dev, rho_old_dev, sc);
        else
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
        std::swap(rho_old_dev, rho_new_dev);
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
        dfree(sc);
    }
};
