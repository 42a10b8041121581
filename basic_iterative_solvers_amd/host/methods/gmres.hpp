// gmres.hpp -- restarted, left-preconditioned GMRES(m), reference
// methods/gmres.hpp.  N-length work (SpMV, preconditioner, the modified
// Gram-Schmidt dots/axpys, the basis combination at a restart) runs on the
// device; the (m+1)x(m+1) Givens/least-squares algebra stays on the host as in
// the reference (gmres.hpp:55-148).  V is one device buffer of (m+1)*N doubles,
// basis vector j at offset j*N (pointer arithmetic as at gmres.hpp:169).
// Restart semantics: Vy = sum_{k<n} y_k v_k (the reference reads y[n] one past
// the end there, SURVEY.md section 5 defect 1; the defined value is 0).
#pragma once

#include "../solver.hpp"

inline void orthogonalize_V(Timers *timers, int N, int n, int m, double *H, double *V, double *w) {
    for (int j = 0; j <= n; ++j) { // modified Gram-Schmidt, gmres.hpp:10-27
        double hjn;
        TIME(timers, "dot", hjn = dot(w, &V[(long)j * N], N))
        H[n + j * m] = hjn;
        TIME(timers, "sum", subtract_vectors(w, w, &V[(long)j * N], N, hjn))
    }
    double hn1;
    TIME(timers, "norm", hn1 = euclidean_vec_norm(w, N))
    H[(n + 1) * m + n] = hn1;
    TIME(timers, "scale", scale(&V[(long)(n + 1) * N], w, 1.0 / hn1, N))
}

// The same modified Gram-Schmidt step with its coefficients kept on the device: j+2 stream-ordered reductions, the
// axpys read their factor from device memory, ONE download of the Hessenberg column (the host-side Givens algebra needs
// it) instead of j+2 blocking dots.  Same kernels, same operation order: H and V come out bit-identical.
inline void orthogonalize_V_dev(Timers *, int N, int n, int m, double *H, double *V, double *w, double *hcol_dev,
                                double *hcol_host) {
    // h_0 = (w, v_0); then one pass per step: w -= h_j v_j together with h_{j+1} = (w, v_{j+1}) -- and the last axpy
    // together with the sum of squares (32 instead of 40 bytes per row and step, one launch pair instead of three)
    dot_dev(w, &V[0], N, hcol_dev);                                          // gmres.hpp:13-14
    for (int j = 0; j < n; ++j)
        axpy_dot_dev(w, &V[(long)j * N], hcol_dev + j, &V[(long)(j + 1) * N], N, hcol_dev + j + 1); // :25, :13-14
    axpy_dot_dev(w, &V[(long)n * N], hcol_dev + n, nullptr, N, hcol_dev + n + 1); // :25, :36-38 (sum of squares ...
    scalar_sqrt_inv(hcol_dev + n + 1, hcol_dev + n + 2, hcol_dev + n + 1);   //  ... its root, and 1/root)
    scale_dev(&V[(long)(n + 1) * N], w, hcol_dev + n + 2, N);                // :44-46
    to_host(hcol_host, hcol_dev, n + 2);
    for (int j = 0; j <= n; ++j) H[n + j * m] = hcol_host[j];
    H[(n + 1) * m + n] = hcol_host[n + 1];
}

inline void least_squares(Timers *timers, int n, int m, double *J, double *H, double *H_tmp, double *Q,
                          double *Q_tmp, double *R) {
    init_dense_identity_matrix(J, m + 1, m + 1);
    init_dense_identity_matrix(H_tmp, m + 1, m);
    if (n == 0) copy_dense_matrix(H_tmp, H, m + 1, m);
    else TIME(timers, "dgemm", dgemm_transpose2(Q, H, H_tmp, m + 1, m + 1, m))
    const double a = H_tmp[n * m + n], b = H_tmp[(n + 1) * m + n];
    const double den = std::sqrt(std::pow(a, 2) + std::pow(b, 2));
    const double c_i = a / den, s_i = b / den;
    J[n * (m + 1) + n] = c_i;
    J[n * (m + 1) + (n + 1)] = s_i;
    J[(n + 1) * (m + 1) + n] = -1.0 * s_i;
    J[(n + 1) * (m + 1) + (n + 1)] = c_i;
    TIME(timers, "dgemm", dgemm_transpose2(J, Q, Q_tmp, m + 1, m + 1, m + 1))
    copy_dense_matrix(Q, Q_tmp, m + 1, m + 1);
    TIME(timers, "dgemm", dgemm_transpose2(Q, H, R, m + 1, m + 1, m))
}

inline void update_g(Timers *timers, int n, int m, double *Q, double *g, double *g_tmp,
                     double &residual_norm, double beta) {
    std::fill(g_tmp, g_tmp + m + 1, 0.0);
    g_tmp[0] = beta;
    std::copy(g_tmp, g_tmp + m + 1, g);
    TIME(timers, "dgemv", dgemv(Q, g, g_tmp, m + 1, m + 1))
    std::copy(g_tmp, g_tmp + m + 1, g);
    residual_norm = std::abs(g[n + 1]);
}

class GMRESSolver : public Solver {
  public:
    double *x = nullptr, *x_old = nullptr, *V = nullptr, *Vy = nullptr, *w = nullptr; // device
    double *y = nullptr, *H = nullptr, *H_tmp = nullptr, *J = nullptr, *Q = nullptr, *Q_tmp = nullptr,
           *R = nullptr, *g = nullptr, *g_tmp = nullptr; // host
    double beta = 0.0;
    bool dev_scalars = true;
    double *hcol_dev = nullptr, *hcol_host = nullptr; // one Hessenberg column: h_0..h_n, ||w||, 1/||w||

    explicit GMRESSolver(const Args *a) : Solver(a), dev_scalars(!a->host_scalars) {}
    void allocate_structs(const int n) override {
        Solver::allocate_structs(n);
        const int m = gmres_restart_len;
        x = dalloc(n); x_old = dalloc(n); V = dalloc((long)n * (m + 1)); Vy = dalloc(n); w = dalloc(n);
        y = new double[m];
        hcol_dev = dalloc(m + 4);
        hcol_host = new double[m + 4];
        H = new double[(m + 1) * m]; H_tmp = new double[(m + 1) * m]; R = new double[(m + 1) * m];
        J = new double[(m + 1) * (m + 1)]; Q = new double[(m + 1) * (m + 1)]; Q_tmp = new double[(m + 1) * (m + 1)];
        g = new double[m + 1]; g_tmp = new double[m + 1];
    }
    void init_structs(const int n) override {
        Solver::init_structs(n);
        const int m = gmres_restart_len;
        if (!gmres_restarted) { copy_vector(x, x_0, n); copy_vector(x_old, x_0, n); }
        init_vector(V, 0.0, (long)n * (m + 1));
        init_vector(Vy, 0.0, n);
        init_vector(w, 0.0, n);
        std::fill(y, y + m, 0.0);
        std::fill(g, g + m + 1, 0.0);
        std::fill(g_tmp, g_tmp + m + 1, 0.0);
        std::fill(H, H + (m + 1) * m, 0.0);
        std::fill(H_tmp, H_tmp + (m + 1) * m, 0.0);
        std::fill(R, R + (m + 1) * m, 0.0);
        init_dense_identity_matrix(J, m + 1, m + 1);
        init_dense_identity_matrix(Q, m + 1, m + 1);
        init_dense_identity_matrix(Q_tmp, m + 1, m + 1);
    }
    void init_residual() override {
        compute_residual(A.get(), x, b, residual, tmp);
        if (!gmres_restarted) { // the unpreconditioned norm opens the history (gmres.hpp:279-284)
            residual_norm = euclidean_vec_norm(residual, N);
            collected_residual_norms[collected_residual_norms_count++] = residual_norm;
        }
        apply_preconditioner(preconditioner, N, L_strict.get(), U_strict.get(), A_D, A_D_inv, L_D, U_D,
                             residual, residual, tmp, work);
        const double pnorm = euclidean_vec_norm(residual, N);
        beta = pnorm;
        g[0] = beta;
        g_tmp[0] = beta;
        scale(V, residual, 1.0 / beta, N);
        if (gmres_restarted) {
            residual_norm = pnorm;
            Solver::init_residual();
        }
    }
    void iterate(Timers *timers) override {
        const int m = gmres_restart_len;
        const int n = iter_count - gmres_restart_count * m;
        TIME(timers, "spmv", spmv(A.get(), &V[(long)n * N], w))
        TIME(timers, "precond", apply_preconditioner(preconditioner, N, L_strict.get(), U_strict.get(), A_D,
                                                     A_D_inv, L_D, U_D, w, w, tmp, work))
        if (dev_scalars) { TIME(timers, "orthog", orthogonalize_V_dev(timers, N, n, m, H, V, w, hcol_dev, hcol_host)) }
        else { TIME(timers, "orthog", orthogonalize_V(timers, N, n, m, H, V, w)) }
        TIME(timers, "least_sq", least_squares(timers, n, m, J, H, H_tmp, Q, Q_tmp, R))
        TIME(timers, "update_g", update_g(timers, n, m, Q, g, g_tmp, residual_norm, beta))
    }
    void get_explicit_x() override {
        const int m = gmres_restart_len;
        const int n = iter_count - gmres_restart_count * m;
        double diag = 1.0;
        for (int r = n - 1; r >= 0; --r) { // back substitution on R, gmres.hpp:337-352
            double sum = 0.0;
            for (int c = r; c < n; ++c) {
                if (r == c) diag = R[r * m + c];
                else sum += R[r * m + c] * y[c];
            }
            y[r] = (g[r] - sum) / diag;
        }
        multi_axpy(V, y, Vy, N, n);
        sum_vectors(x, x_old, Vy, N);
    }
    void save_x_star() override {
        get_explicit_x();
        std::swap(x, x_star);
        Solver::save_x_star();
    }
    void check_restart(Timers *timers) override {
        const bool conv = residual_norm < stopping_criteria;
        const bool over = iter_count > max_iters;
        const bool cycle = (iter_count % gmres_restart_len == 0) && iter_count != 0;
        if (!conv && !over && cycle) {
            gmres_restarted = true;
            get_explicit_x();
            copy_vector(x_old, x, N);
            init_structs(N);
            init_residual();
            time_per_iteration[collected_residual_norms_count] = timers->per_iteration_time->check();
            ++gmres_restart_count;
        }
    }
    void exchange() override {}
    ~GMRESSolver() override {
        dfree(x); dfree(x_old); dfree(V); dfree(Vy); dfree(w); dfree(hcol_dev);
        delete[] hcol_host;
        delete[] y; delete[] H; delete[] H_tmp; delete[] J; delete[] Q; delete[] Q_tmp; delete[] R;
        delete[] g; delete[] g_tmp;
    }
};
