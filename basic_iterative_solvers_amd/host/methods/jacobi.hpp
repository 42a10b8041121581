// jacobi.hpp -- Jacobi iteration, reference methods/jacobi.hpp:43-52 (iteration),
// :79-107 (residual bookkeeping: a full true residual every iteration).
// Two schedules with identical arithmetic: the reference's kernel-by-kernel order with a blocking norm
// per iteration (-unfused), and the device schedule bis_stat_* (default): one SpMV per iteration -- the
// product that samples iteration k's residual is the one iteration k+1 starts from -- with the norm and
// the stopping test on the device; the sampled norms are bit-identical between the two.
#pragma once

#include "../solver.hpp"

inline void jacobi_separate_iteration(Timers *timers, const MatrixCRS *A, const double *D,
                                      const double *b, double *x_new, const double *x_old) {
    TIME(timers, "spmv", spmv(A, x_old, x_new))                         // x_new <- A x_old
    TIME(timers, "normalize", normalize_x(x_new, x_old, D, b, A->n_rows)) // x_new <- (b-(x_new-D x_old))/D
}

// enqueue the device schedule in chunks of `chunk` iterations until its stop test has fired (or max_iters), then
// hand the residual history to the harness, which replays it iteration by iteration
inline void run_stat_schedule(Timers *timers, bis_stat *st, int max_iters, int chunk, std::vector<double> &hist) {
    int it = 0, conv = 0, enq = 0;
    hist.assign(1, 0.0);
    while (enq < max_iters) {
        const int batch = std::min(chunk, max_iters - enq);
        TIME(timers, "spmv", bis::check(bis_stat_iterate(bis::ctx(), st, batch), "bis_stat_iterate"))
        enq += batch;
        hist.resize(enq + 1);
        bis::check(bis_stat_status(bis::ctx(), st, &it, &conv, hist.data(), enq + 1), "bis_stat_status");
        if (it < enq) break; // the device stopped inside this chunk
    }
    hist.resize(it + 1);
}

class JacobiSolver : public Solver {
  public:
    double *x_new = nullptr, *x_old = nullptr;
    bool fused = false;
    bis_stat *fst = nullptr;
    std::vector<double> fused_hist;
    explicit JacobiSolver(const Args *a) : Solver(a) { fused = !a->unfused && residual_check_len == 1; }
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
        if (fused) {
            bis::check(bis_stat_create(bis::ctx(), BIS_STAT_JACOBI, A->dev, nullptr, nullptr, A_D, b, x_old, &fst), "bis_stat_create");
            bis::check(bis_stat_init(bis::ctx(), fst, tolerance, &residual_norm), "bis_stat_init");
            collected_residual_norms[collected_residual_norms_count++] = residual_norm;
            return;
        }
        compute_residual(A.get(), x_old, b, residual, tmp);
        residual_norm = euclidean_vec_norm(residual, N);
        Solver::init_residual();
    }
    void iterate(Timers *timers) override {
        if (fused) { // the whole solve at the first call, in chunks with a status read in between (cg.hpp does the same)
            if (fused_hist.empty()) run_stat_schedule(timers, fst, max_iters, 50, fused_hist);
            return;
        }
        jacobi_separate_iteration(timers, A.get(), A_D, b, x_new, x_old);
    }
    void exchange() override { if (!fused) std::swap(x_old, x_new); }
    void save_x_star() override {
        if (fused) bis::check(bis_stat_solution(bis::ctx(), fst, x_old), "bis_stat_solution");
        std::swap(x_old, x_star);
        Solver::save_x_star();
    }
    void record_residual_norm() override {
        if (fused) {
            residual_norm = fused_hist[std::min<size_t>(iter_count, fused_hist.size() - 1)];
        } else {
            compute_residual(A.get(), x_new, b, residual, tmp);
            residual_norm = euclidean_vec_norm(residual, N);
        }
        Solver::record_residual_norm();
    }
    ~JacobiSolver() override {
        if (fst) bis_stat_destroy(bis::ctx(), fst);
        dfree(x_new); dfree(x_old);
    }
};
