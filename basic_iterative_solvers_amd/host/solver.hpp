// solver.hpp -- abstract Solver of the MI355X build: same fields, virtuals and
// bookkeeping as the reference's Solver (solver.hpp:9-193); every N-vector is
// a device buffer and every loop over N is a kernel call.
#pragma once

#include <cfloat>
#include <cstring>

#include "common.hpp"
#include "kernels.hpp"
#include "sparse_matrix.hpp"

class Solver {
  public:
    SolverType method;
    PrecondType preconditioner = PrecondType::None;
    std::unique_ptr<MatrixCRS> A, L, L_strict, U, U_strict;

    double stopping_criteria = 0.0;
    int iter_count = 0;
    int collected_residual_norms_count = 0;
    double residual_norm = DBL_MAX;
    int max_iters = MAX_ITERS;
    double tolerance = TOL;
    int residual_check_len = RES_CHECK_LEN;
    int gmres_restart_len = 0;
    int gmres_restart_count = 0;
    bool num_scale = false;
    int N = 0;

    // device vectors (solver.hpp:37-48)
    double *x_star = nullptr, *x_0 = nullptr, *b = nullptr, *tmp = nullptr, *work = nullptr,
           *residual = nullptr, *residual_0 = nullptr, *A_D = nullptr, *A_D_inv = nullptr,
           *A_D_scale = nullptr, *L_D = nullptr, *U_D = nullptr;

    double *collected_residual_norms = nullptr; // host
    double *time_per_iteration = nullptr;       // host
    bool convergence_flag = false;
    bool gmres_restarted = false;
    // -perm: perm[new] = old as int32 in device storage (nullptr: natural order).  The solve runs on
    // P A P^T; unpermute_x_star() returns x* in the caller's row order.
    double *perm_store = nullptr;
    void keep_permutation(const std::vector<int> &perm) {
        const long n = (long)perm.size();
        std::vector<double> raw((n + 1) / 2 + 1, 0.0);
        std::memcpy(raw.data(), perm.data(), sizeof(int) * (size_t)n);
        if (!perm_store) perm_store = dalloc((n + 1) / 2 + 1);
        to_device(perm_store, raw.data(), (n + 1) / 2 + 1);
    }
    void unpermute_x_star() {
        if (!perm_store) return;
        bis::check(bis_vec_scatter(bis::ctx(), tmp, x_star, reinterpret_cast<const int32_t *>(perm_store), N), "bis_vec_scatter");
        copy_vector(x_star, tmp, N);
    }

    explicit Solver(const Args *a)
        : method(a->method), preconditioner(a->preconditioner), gmres_restart_len(a->restart_length),
          num_scale(a->num_scale) {
        collected_residual_norms = new double[max_iters * 2]();
        time_per_iteration = new double[max_iters * 2]();
    }

    virtual void iterate(Timers *) = 0;
    virtual void exchange() = 0;

    virtual void allocate_structs(const int n) {
        N = n;
        double **v[] = {&x_star, &x_0, &b, &tmp, &work, &residual, &residual_0, &A_D, &A_D_inv,
                        &A_D_scale, &L_D, &U_D};
        for (auto p : v) *p = dalloc(n);
        if (!gmres_restarted) { // defaults of solver.hpp:96-108
            init_vector(x_star, 0.0, n);
            init_vector(x_0, INIT_X_VAL, n);
            init_vector(b, B_VAL, n);
            init_vector(A_D, 1.0, n);
            init_vector(A_D_inv, 0.0, n);
            init_vector(A_D_scale, 0.0, n);
            init_vector(L_D, 1.0, n);
            init_vector(U_D, 1.0, n);
        }
    }
    virtual void init_structs(const int n) {
        init_vector(tmp, 0.0, n);
        init_vector(work, 0.0, n);
        init_vector(residual, 0.0, n);
        init_vector(residual_0, 0.0, n);
    }
    virtual void check_restart(Timers *) {}
    virtual void get_explicit_x() {}
    virtual ~Solver() {
        double *v[] = {x_star, x_0, b, tmp, work, residual, residual_0, A_D, A_D_inv, A_D_scale, L_D, U_D};
        for (auto p : v) dfree(p);
        dfree(perm_store);
        delete[] collected_residual_norms;
        delete[] time_per_iteration;
    }

    virtual void init_residual() {
        copy_vector(residual_0, residual, N);
        collected_residual_norms[collected_residual_norms_count++] = residual_norm;
    }
    virtual void save_x_star() {
        compute_residual(A.get(), x_star, b, residual, tmp);
        residual_norm = euclidean_vec_norm(residual, N);
        collected_residual_norms[collected_residual_norms_count + 1] = residual_norm;
    }
    virtual void record_residual_norm() { collected_residual_norms[collected_residual_norms_count++] = residual_norm; }

    void sample_residual(Stopwatch *per_iteration_time) {
        if (iter_count % residual_check_len == 0) {
            record_residual_norm();
            time_per_iteration[collected_residual_norms_count] = per_iteration_time->check();
        }
    }
    void init_stopping_criteria() { stopping_criteria = tolerance * residual_norm; }
    bool check_stopping_criteria() {
        const bool norm_convergence = std::abs(residual_norm) < stopping_criteria;
        const bool over_max_iters = iter_count >= (max_iters - gmres_restart_count);
        const bool divergence = std::abs(residual_norm) > DBL_MAX || std::isnan(residual_norm);
        return norm_convergence || over_max_iters || divergence;
    }
};
