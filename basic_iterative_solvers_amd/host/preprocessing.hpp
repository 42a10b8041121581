// preprocessing.hpp -- setup phase, reference preprocessing.hpp:26-100:
// allocate + initialise the solver's vectors, optional symmetric diagonal
// scaling, split A into its strict triangles and diagonal (on the device for
// generated matrices, on the host for file input), optional ILU(0), initial
// residual and stopping criterion.
#pragma once

#include "common.hpp"
#include "solver.hpp"
#include "utilities/LU_factors.hpp"
#include "utilities/permute.hpp"

inline void download_to_host(MatrixCRS *A) { // device-generated matrix -> host arrays (setup only)
    if (A->row_ptr) return;
    A->row_ptr = new crs_index[A->n_rows + 1];
    A->col = new int[A->nnz ? A->nnz : 1];
    A->val = new double[A->nnz ? A->nnz : 1];
    bis::check(bis_mat_download(bis::ctx(), A->dev, A->row_ptr, A->col, A->val), "bis_mat_download");
}

inline void scale_mat(MatrixCRS *A, const double *s) { // preprocessing.hpp:15-24
    for (int r = 0; r < A->n_rows; ++r)
        for (crs_index i = A->row_ptr[r]; i < A->row_ptr[r + 1]; ++i) A->val[i] *= (s[r] * s[A->col[i]]);
}

// factor_LU, utilities/LU_factors.hpp:900-934
inline void factor_LU(Solver *s) {
    // split + diagonal on the device (bis_mat_split_strict: bit-identical to the reference's split_LU + peel_diag_crs,
    // tests/test_gpu_kernels.py), for generated and for file inputs alike; the host versions of utilities/LU_factors.hpp
    // remain for callers that hold host matrices
    {
        bis_mat *Ls = nullptr, *Us = nullptr;
        const bis_status st = bis_mat_split_strict(bis::ctx(), s->A->dev, &Ls, &Us, s->A_D, s->A_D_inv);
        if (st == BIS_ERR_ZERO_DIAG || st == BIS_ERR_NO_DIAG) { fprintf(stderr, "%s\n", bis_last_error(bis::ctx())); exit(EXIT_FAILURE); }
        bis::check(st, "bis_mat_split_strict");
        s->L_strict->adopt(Ls);
        s->U_strict->adopt(Us);
    }
    if (s->preconditioner == PrecondType::ILU0) {
        // The reference's wired-in factor_ILU0_new needs the SMAX library (SURVEY.md
        // section 5, defect 2); this is its serial factor_ILU0_old arithmetic,
        // level-scheduled on the device.  Overwrites L_strict/U_strict, L_D, U_D.
        bis_mat *Ls = nullptr, *Us = nullptr;
        bis::check(bis_mat_ilu0(bis::ctx(), s->A->dev, ILU0_PIVOT_TOLERANCE, ILU0_PIVOT_REPLACEMENT, &Ls, &Us,
                                s->L_D, s->U_D), "bis_mat_ilu0");
        s->L_strict->free_host();
        s->U_strict->free_host();
        s->L_strict->adopt(Ls);
        s->U_strict->adopt(Us);
    }
}

inline void preprocessing(Args *cli_args, Solver *solver, Timers *timers, std::unique_ptr<MatrixCRS> &A) {
    (*timers)["preprocessing_init"].start();
    solver->allocate_structs(A->n_cols);
    solver->init_structs(A->n_cols);
    (*timers)["preprocessing_init"].stop();
    solver->A = std::move(A);

    if (solver->num_scale) { // preprocessing.hpp:39-50: A' = D^-1/2 A D^-1/2, b' = D^-1/2 b
        const int N = solver->A->n_rows;
        // extract_scale + scale_mat on the device (bis_mat_scale_sym); a matrix read from a file keeps its
        // host copy in step so that later host-side steps (host reordering fallback) see the scaled values
        init_vector(solver->A_D_scale, 0.0, N);
        {
            const bis_status sst = bis_mat_scale_sym(bis::ctx(), solver->A->dev, solver->A_D_scale);
            if (sst == BIS_ERR_ZERO_DIAG) { fprintf(stderr, "%s\n", bis_last_error(bis::ctx())); exit(EXIT_FAILURE); }
            bis::check(sst, "bis_mat_scale_sym");
        }
        if (solver->A->row_ptr) { // host copy present
            std::vector<double> s(N, 0.0);
            to_host(s.data(), solver->A_D_scale, N);
            scale_mat(solver->A.get(), s.data());
        }
        // the reference also scales x_0 here, after init_structs has already
        // copied the unscaled x_0 into the iterate; x_0 is not read again
        elemwise_mult_vectors(solver->x_0, solver->A_D_scale, solver->x_0, N);
        elemwise_mult_vectors(solver->b, solver->A_D_scale, solver->b, N);
    }

    if (cli_args->perm_mode == "mc") { // the SMAX permute_mat step (preprocessing.hpp:58-62)
        const int N = solver->A->n_rows;
        int n_colours = 0;
        std::vector<int> perm;
        // device path: colouring, permutation and P A P^T stay in HBM
        double *perm_store = nullptr;
        bis::check(bis_vec_alloc(bis::ctx(), (N + 1) / 2 + 1, &perm_store), "bis_vec_alloc");
        int32_t *perm_dev = reinterpret_cast<int32_t *>(perm_store);
        bis_mat *Bm = nullptr;
        const bis_status mc = cli_args->perm_host ? BIS_ERR_UNSUPPORTED
                                                  : bis_mat_multicolour(bis::ctx(), solver->A->dev, &Bm, perm_dev, &n_colours);
        if (mc == BIS_OK) {
            auto B = std::make_unique<MatrixCRS>();
            B->adopt(Bm);
            solver->A = std::move(B);
            if (solver->num_scale) { // b was rescaled row-wise: permute it (x_0 is constant)
                double *pb = nullptr;
                bis::check(bis_vec_alloc(bis::ctx(), N, &pb), "bis_vec_alloc");
                bis::check(bis_vec_gather(bis::ctx(), pb, solver->b, perm_dev, N), "bis_vec_gather");
                copy_vector(solver->b, pb, N);
                bis::check(bis_vec_free(bis::ctx(), pb), "bis_vec_free");
            }
            if (!cli_args->dump_perm.empty()) {
                std::vector<double> raw((N + 1) / 2 + 1);
                to_host(raw.data(), perm_store, (N + 1) / 2 + 1);
                const int32_t *pp = reinterpret_cast<const int32_t *>(raw.data());
                perm.assign(pp, pp + N);
            }
        } else { // more than 64 colours (or -perm-host): the host version
            download_to_host(solver->A.get());
            std::vector<int> inv_perm;
            multicolour_permutation(solver->A.get(), perm, inv_perm, n_colours);
            auto B = std::make_unique<MatrixCRS>();
            permute_matrix(solver->A.get(), perm, inv_perm, B.get());
            B->upload();
            solver->A = std::move(B);
            if (solver->num_scale) {
                std::vector<double> hb(N), pb(N);
                to_host(hb.data(), solver->b, N);
                for (int i = 0; i < N; ++i) pb[i] = hb[perm[i]];
                to_device(solver->b, pb.data(), N);
            }
        }
        if (mc == BIS_OK) solver->perm_store = perm_store; // kept: x* goes back to the natural order after the solve
        else { bis::check(bis_vec_free(bis::ctx(), perm_store), "bis_vec_free"); solver->keep_permutation(perm); }
        if (!cli_args->dump_perm.empty()) write_permutation(cli_args->dump_perm, perm);
        std::cout << "multi-colour reordering: " << n_colours << " colours" << std::endl;
    } else if (cli_args->perm_mode == "rcm" || cli_args->perm_mode == "bfs") {
        // bandwidth-reducing orderings: they keep the sweeps level-scheduled, with fewer or more levels than the
        // natural order depending on the input.  On the device (level-synchronous BFS, bis_order.hip) for structurally
        // symmetric patterns; the sequential host version otherwise, or with -perm-host.
        const int N = solver->A->n_rows;
        const bool rcm = cli_args->perm_mode == "rcm";
        std::vector<int> perm;
        double *perm_store = dalloc((N + 1) / 2 + 1);
        int32_t *perm_dev = reinterpret_cast<int32_t *>(perm_store);
        bis_status dev = cli_args->perm_host ? BIS_ERR_UNSUPPORTED : bis_mat_bfs_order(bis::ctx(), solver->A->dev, rcm ? 1 : 0, perm_dev);
        if (dev != BIS_OK && dev != BIS_ERR_UNSUPPORTED) bis::check(dev, "bis_mat_bfs_order");
        if (dev == BIS_OK) {
            bis_mat *Bm = nullptr;
            bis::check(bis_mat_permute(bis::ctx(), solver->A->dev, perm_dev, &Bm), "bis_mat_permute");
            auto B = std::make_unique<MatrixCRS>();
            B->adopt(Bm);
            solver->A = std::move(B);
            if (solver->num_scale) { // b was rescaled row-wise: permute it (x_0 is constant)
                double *pb = dalloc(N);
                bis::check(bis_vec_gather(bis::ctx(), pb, solver->b, perm_dev, N), "bis_vec_gather");
                copy_vector(solver->b, pb, N);
                dfree(pb);
            }
            solver->perm_store = perm_store;
            if (!cli_args->dump_perm.empty()) {
                std::vector<double> raw((N + 1) / 2 + 1);
                to_host(raw.data(), perm_store, (N + 1) / 2 + 1);
                const int32_t *pp = reinterpret_cast<const int32_t *>(raw.data());
                perm.assign(pp, pp + N);
            }
        } else {
            dfree(perm_store);
            download_to_host(solver->A.get());
            std::vector<int> inv_perm;
            bfs_like_permutation(solver->A.get(), rcm, perm, inv_perm);
            auto B = std::make_unique<MatrixCRS>();
            permute_matrix(solver->A.get(), perm, inv_perm, B.get());
            B->upload();
            solver->A = std::move(B);
            if (solver->num_scale) {
                std::vector<double> hb(N), pb(N);
                to_host(hb.data(), solver->b, N);
                for (int i = 0; i < N; ++i) pb[i] = hb[perm[i]];
                to_device(solver->b, pb.data(), N);
            }
            solver->keep_permutation(perm);
        }
        if (!cli_args->dump_perm.empty()) write_permutation(cli_args->dump_perm, perm);
        std::cout << (rcm ? "reverse Cuthill-McKee" : "breadth-first") << " reordering" << (dev == BIS_OK ? "" : " (host)") << std::endl;
    } else if (cli_args->perm_mode != "none") {
        fprintf(stderr, "ERROR: unknown -perm mode (available: mc, rcm, bfs)\n");
        exit(EXIT_FAILURE);
    }

    solver->L = std::make_unique<MatrixCRS>();
    solver->L_strict = std::make_unique<MatrixCRS>();
    solver->U = std::make_unique<MatrixCRS>();
    solver->U_strict = std::make_unique<MatrixCRS>();
    TIME(timers, "preprocessing_factor", factor_LU(solver))

    (*timers)["preprocessing_init"].start();
    solver->init_residual();
    solver->init_stopping_criteria();
    (*timers)["preprocessing_init"].stop();
}
