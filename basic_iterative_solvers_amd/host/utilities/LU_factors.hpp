// LU_factors.hpp -- host-side setup (run once): A -> L, L_strict, U, U_strict,
// and diagonal extraction, following the reference's
// utilities/LU_factors.hpp (split_LU :122-309, peel_diag_crs :827-869,
// extract_scale :880-898, factor_LU :900-934); ILU(0) itself runs on the device
// (bis_mat_ilu0).
// Outputs are bit-identical CRS arrays; the strict parts are uploaded to the
// device where the triangular solves run.
#pragma once

#include "../common.hpp"
#include "../sparse_matrix.hpp"

inline void split_LU(const MatrixCRS *A, MatrixCRS *L, MatrixCRS *L_strict, MatrixCRS *U,
                     MatrixCRS *U_strict) {
    const int n = A->n_rows;
    long cnt[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (crs_index k = A->row_ptr[i]; k < A->row_ptr[i + 1]; ++k) {
            const int c = A->col[k];
            if (c < i) { ++cnt[0]; ++cnt[1]; }
            if (c == i) { ++cnt[0]; ++cnt[2]; }
            if (c > i) { ++cnt[2]; ++cnt[3]; }
        }
    MatrixCRS *out[4] = {L, L_strict, U, U_strict};
    for (int m = 0; m < 4; ++m) {
        out[m]->free_host();
        out[m]->n_rows = n; out[m]->n_cols = A->n_cols; out[m]->nnz = (crs_index)cnt[m];
        out[m]->row_ptr = new crs_index[n + 1];
        out[m]->col = new int[cnt[m] ? cnt[m] : 1];
        out[m]->val = new double[cnt[m] ? cnt[m] : 1];
        out[m]->row_ptr[0] = 0;
    }
    crs_index p[4] = {0, 0, 0, 0};
    auto put = [&](int m, int c, double v) { out[m]->col[p[m]] = c; out[m]->val[p[m]++] = v; };
    for (int i = 0; i < n; ++i) {
        for (crs_index k = A->row_ptr[i]; k < A->row_ptr[i + 1]; ++k) {
            const int c = A->col[k];
            const double v = A->val[k];
            if (c < i) { put(0, c, v); put(1, c, v); }
            if (c == i) { put(0, c, v); put(2, c, v); }
            if (c > i) { put(2, c, v); put(3, c, v); }
        }
        for (int m = 0; m < 4; ++m) out[m]->row_ptr[i + 1] = p[m];
    }
}

// D (and 1/D) from the diagonal; the diagonal entry is swapped to the row end.
inline void peel_diag_crs(MatrixCRS *A, double *D, double *D_inv = nullptr) {
    for (int r = 0; r < A->n_rows; ++r) {
        const crs_index start = A->row_ptr[r], last = A->row_ptr[r + 1] - 1;
        crs_index dj = -1;
        for (crs_index j = start; j <= last; ++j)
            if (A->col[j] == r) {
                dj = j;
                D[r] = A->val[j];
                if (std::abs(D[r]) < 1e-16) SanityChecker::zero_diag(r);
                if (D_inv) D_inv[r] = 1.0 / D[r];
            }
        if (dj < 0) SanityChecker::no_diag(r);
        if (dj != last) { std::swap(A->col[dj], A->col[last]); std::swap(A->val[dj], A->val[last]); }
    }
}

inline void extract_scale(MatrixCRS *A, double *D_scale) {
    for (int r = 0; r < A->n_rows; ++r)
        for (crs_index j = A->row_ptr[r]; j < A->row_ptr[r + 1]; ++j)
            if (A->col[j] == r) {
                if (std::abs(A->val[j]) < 1e-16) SanityChecker::zero_diag(r);
                D_scale[r] = 1.0 / std::sqrt(std::abs(A->val[j]));
            }
}
