// permute.hpp -- opt-in symmetric reordering for parallel triangular sweeps
// (the role SMAX's PERM_MODE plays in the reference: CMakeLists.txt:128-133,
// utilities/smax_helpers.hpp:44-80).  `-perm mc`: greedy multi-colouring of the
// symmetrised pattern, rows grouped by colour (stable), A <- P A P^T.  Rows of
// one colour do not couple, so the strict triangles of the permuted matrix
// have as many dependency levels as there are colours (2 for a 7-point, 8 for
// a 27-point stencil) instead of 3n-2 / 7n-6.  This CHANGES the Gauss-Seidel /
// ILU iteration (a different preconditioner); its parity target is the
// reference run on the same permuted matrix (tests/test_host_cli.py).
#pragma once

#include <algorithm>
#include <fstream>
#include <numeric>

#include "../sparse_matrix.hpp"

// perm[new] = old, inv_perm[old] = new
inline void multicolour_permutation(const MatrixCRS *A, std::vector<int> &perm, std::vector<int> &inv_perm,
                                    int &n_colours) {
    const int n = A->n_rows;
    // symmetrised adjacency (pattern of A + A^T) in CSR form
    std::vector<int> deg(n + 1, 0);
    for (int r = 0; r < n; ++r)
        for (int k = A->row_ptr[r]; k < A->row_ptr[r + 1]; ++k) {
            const int c = A->col[k];
            if (c != r) { ++deg[r + 1]; ++deg[c + 1]; }
        }
    for (int r = 0; r < n; ++r) deg[r + 1] += deg[r];
    std::vector<int> adj(deg[n]), fill(deg.begin(), deg.end() - 1);
    for (int r = 0; r < n; ++r)
        for (int k = A->row_ptr[r]; k < A->row_ptr[r + 1]; ++k) {
            const int c = A->col[k];
            if (c != r) { adj[fill[r]++] = c; adj[fill[c]++] = r; }
        }
    std::vector<int> colour(n, -1), mark;
    n_colours = 0;
    for (int r = 0; r < n; ++r) { // first-fit in natural order
        mark.assign(n_colours + 1, 0);
        for (int k = deg[r]; k < deg[r + 1]; ++k)
            if (colour[adj[k]] >= 0) mark[colour[adj[k]]] = 1;
        int c = 0;
        while (c < n_colours && mark[c]) ++c;
        colour[r] = c;
        if (c == n_colours) ++n_colours;
    }
    perm.resize(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return colour[a] < colour[b]; });
    inv_perm.resize(n);
    for (int i = 0; i < n; ++i) inv_perm[perm[i]] = i;
}

// B = P A P^T: new row i is old row perm[i]; entries keep their order inside
// the row, columns are renumbered through inv_perm.
inline void permute_matrix(const MatrixCRS *A, const std::vector<int> &perm, const std::vector<int> &inv_perm,
                           MatrixCRS *B) {
    const int n = A->n_rows;
    B->free_host();
    B->n_rows = n; B->n_cols = A->n_cols; B->nnz = A->nnz;
    B->row_ptr = new int[n + 1];
    B->col = new int[A->nnz ? A->nnz : 1];
    B->val = new double[A->nnz ? A->nnz : 1];
    B->row_ptr[0] = 0;
    for (int i = 0; i < n; ++i) B->row_ptr[i + 1] = B->row_ptr[i] + (A->row_ptr[perm[i] + 1] - A->row_ptr[perm[i]]);
    for (int i = 0; i < n; ++i) {
        int p = B->row_ptr[i];
        for (int k = A->row_ptr[perm[i]]; k < A->row_ptr[perm[i] + 1]; ++k) {
            B->col[p] = inv_perm[A->col[k]];
            B->val[p++] = A->val[k];
        }
    }
}

inline void write_permutation(const std::string &path, const std::vector<int> &perm) {
    std::ofstream f(path);
    for (int v : perm) f << v << "\n";
}
