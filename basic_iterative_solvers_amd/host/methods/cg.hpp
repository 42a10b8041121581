// cg.hpp -- (preconditioned) conjugate gradient, reference methods/cg.hpp.
// Two schedules with identical arithmetic:
//   * cg_separate_iteration: the reference's kernel-by-kernel order
//     (cg.hpp:6-54), three blocking dots per iteration;
//   * the fused device schedule (bis_cg_*: three streaming passes, alpha/beta
//     and the stop test on the device) -- the default; -p none / -p j inside its
//     update pass, every other preconditioner through bis_cg_set_preconditioner.
#pragma once

#include "../solver.hpp"

inline void cg_separate_iteration(Timers *timers, const PrecondType pc, const MatrixCRS *A,
                                  const MatrixCRS *L, const MatrixCRS *U, double *A_D, double *A_D_inv,
                                  double *L_D, double *U_D, double *x_new, double *x_old, double *tmp,
                                  double *work, double *p_new, double *p_old, double *r_new,
                                  double *r_old, double *z_new, double *z_old) {
    const int N = A->n_cols;
    TIME(timers, "spmv", spmv(A, p_old, tmp))
    double tmp_dot;
    TIME(timers, "dot", tmp_dot = dot(r_old, z_old, N))
    double alpha;
    TIME(timers, "dot", alpha = tmp_dot / dot(tmp, p_old, N))
    TIME(timers, "sum", sum_vectors(x_new, x_old, p_old, N, alpha))
    TIME(timers, "sum", subtract_vectors(r_new, r_old, tmp, N, alpha))
    TIME(timers, "precond", apply_preconditioner(pc, N, L, U, A_D, A_D_inv, L_D, U_D, z_new, r_new, tmp, work))
    double beta;
    TIME(timers, "dot", beta = dot(r_new, z_new, N) / tmp_dot)
    TIME(timers, "sum", sum_vectors(p_new, z_new, p_old, N, beta))
}

class ConjugateGradientSolver : public Solver {
  public:
    double *x_new = nullptr, *x_old = nullptr, *p_old = nullptr, *p_new = nullptr, *z_old = nullptr,
           *z_new = nullptr, *residual_old = nullptr, *residual_new = nullptr;
    bool fused = false;
    bis_cg *fcg = nullptr;
    std::vector<double> fused_hist;

    explicit ConjugateGradientSolver(const Args *a) : Solver(a) {
        fused = !a->unfused; // every preconditioner: None / Jacobi inside pass B, the others through bis_cg_set_preconditioner
    }
    void allocate_structs(const int n) override {
        Solver::allocate_structs(n);
        double **v[] = {&x_new, &x_old, &p_new, &p_old, &residual_new, &residual_old, &z_new, &z_old};
        for (auto p : v) *p = dalloc(n);
    }
    void init_structs(const int n) override {
        Solver::init_structs(n);
        double *z[] = {x_new, p_new, p_old, residual_new, residual_old, z_new, z_old};
        for (auto p : z) init_vector(p, 0.0, n);
        copy_vector(x_old, x_0, n);
    }
    void init_residual() override {
        if (fused) {
            bis::check(bis_cg_create(bis::ctx(), A->dev, preconditioner == PrecondType::Jacobi ? A_D : nullptr,
                                     b, x_old, &fcg), "bis_cg_create");
            if (preconditioner != PrecondType::None && preconditioner != PrecondType::Jacobi)
                bis::check(bis_cg_set_preconditioner(bis::ctx(), fcg, (int)preconditioner, L_strict ? L_strict->dev : nullptr,
                                                     U_strict ? U_strict->dev : nullptr, A_D, A_D_inv, L_D, U_D,
                                                     PRECOND_OUTER_ITERS, PRECOND_INNER_ITERS), "bis_cg_set_preconditioner");
            bis::check(bis_cg_init(bis::ctx(), fcg, tolerance, &residual_norm), "bis_cg_init");
            collected_residual_norms[collected_residual_norms_count++] = residual_norm;
            return;
        }
        compute_residual(A.get(), x_old, b, residual, tmp);
        apply_preconditioner(preconditioner, N, L_strict.get(), U_strict.get(), A_D, A_D_inv, L_D, U_D,
                             z_old, residual, tmp, work);
        copy_vector(p_old, z_old, N);
        copy_vector(residual_old, residual, N);
        residual_norm = euclidean_vec_norm(residual, N);
        Solver::init_residual();
    }
    void iterate(Timers *timers) override {
        if (fused) {
            // enqueue a batch, then read the device-side history; the device stops
            // updating at the reference's stopping iteration (bis_cg.hip)
            if (fused_hist.empty()) {
                // chunks of 50 iterations with a status read in between: passes enqueued
                // behind the stopping iteration are no-ops, but they are still launches -- and the
                // sweeps of a general preconditioner do not see the stop flag at all: chunks of 8 there
                const bool sweeps = preconditioner != PrecondType::None && preconditioner != PrecondType::Jacobi;
                int it = 0, conv = 0, enq = 0;
                fused_hist.assign(1, 0.0);
                while (enq < max_iters) {
                    const int batch = std::min(sweeps ? 8 : 50, max_iters - enq);
                    TIME(timers, "spmv", bis::check(bis_cg_iterate(bis::ctx(), fcg, batch), "bis_cg_iterate"))
                    enq += batch;
                    fused_hist.resize(enq + 1);
                    bis::check(bis_cg_status(bis::ctx(), fcg, &it, &conv, fused_hist.data(), enq + 1), "bis_cg_status");
                    if (it < enq) break; // the device stopped inside this chunk
                }
                fused_hist.resize(it + 1);
            }
            return;
        }
        cg_separate_iteration(timers, preconditioner, A.get(), L_strict.get(), U_strict.get(), A_D, A_D_inv,
                              L_D, U_D, x_new, x_old, tmp, work, p_new, p_old, residual_new, residual_old,
                              z_new, z_old);
    }
    void exchange() override {
        if (fused) return; // the fused schedule updates in place
        std::swap(p_old, p_new);
        std::swap(z_old, z_new);
        std::swap(residual_old, residual_new);
        std::swap(x_old, x_new);
    }
    void save_x_star() override {
        std::swap(x_old, x_star);
        Solver::save_x_star();
    }
    void record_residual_norm() override {
        if (fused) residual_norm = fused_hist[std::min<size_t>(iter_count, fused_hist.size() - 1)];
        else residual_norm = euclidean_vec_norm(residual_new, N);
        Solver::record_residual_norm();
    }
    ~ConjugateGradientSolver() override {
        if (fcg) bis_cg_destroy(bis::ctx(), fcg);
        double *v[] = {x_new, x_old, p_new, p_old, residual_new, residual_old, z_new, z_old};
        for (auto p : v) dfree(p);
    }
};
