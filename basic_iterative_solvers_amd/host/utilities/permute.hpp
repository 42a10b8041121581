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
        for (crs_index k = A->row_ptr[r]; k < A->row_ptr[r + 1]; ++k) {
            const int c = A->col[k];
            if (c != r) { ++deg[r + 1]; ++deg[c + 1]; }
        }
    for (int r = 0; r < n; ++r) deg[r + 1] += deg[r];
    std::vector<int> adj(deg[n]), fill(deg.begin(), deg.end() - 1);
    for (int r = 0; r < n; ++r)
        for (crs_index k = A->row_ptr[r]; k < A->row_ptr[r + 1]; ++k) {
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

// symmetrised adjacency (pattern of A + A^T without the diagonal) in CSR form
inline void symmetric_adjacency(const MatrixCRS *A, std::vector<int> &ptr, std::vector<int> &adj) {
    const int n = A->n_rows;
    ptr.assign(n + 1, 0);
    for (int r = 0; r < n; ++r)
        for (crs_index k = A->row_ptr[r]; k < A->row_ptr[r + 1]; ++k) {
            const int c = A->col[k];
            if (c != r) { ++ptr[r + 1]; ++ptr[c + 1]; }
        }
    for (int r = 0; r < n; ++r) ptr[r + 1] += ptr[r];
    adj.resize(ptr[n]);
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (int r = 0; r < n; ++r)
        for (crs_index k = A->row_ptr[r]; k < A->row_ptr[r + 1]; ++k) {
            const int c = A->col[k];
            if (c != r) { adj[fill[r]++] = c; adj[fill[c]++] = r; }
        }
    // a structurally symmetric entry pair shows up twice: sort + unique per row, then compact
    std::vector<int> cptr(n + 1, 0);
    int w = 0;
    for (int r = 0; r < n; ++r) {
        std::sort(adj.begin() + ptr[r], adj.begin() + ptr[r + 1]);
        const int a = ptr[r], b = ptr[r + 1];
        cptr[r] = w;
        for (int k = a; k < b; ++k)
            if (k == a || adj[k] != adj[k - 1]) adj[w++] = adj[k];
    }
    cptr[n] = w;
    adj.resize(w);
    ptr.swap(cptr);
}

// `-perm bfs`: breadth-first ordering of every connected component from its lowest-numbered
// vertex, neighbours visited in ascending index order.  `-perm rcm`: Cuthill-McKee (start at a
// vertex of minimum degree, neighbours by ascending degree, ties by index), reversed.  Both are
// the roles of SMAX's PERM_MODE BFS / RCM (CMakeLists.txt:128-133); the permutation itself is
// this layer's, the parity target is the reference algorithm on the permuted matrix.
inline void bfs_like_permutation(const MatrixCRS *A, bool rcm, std::vector<int> &perm, std::vector<int> &inv_perm) {
    const int n = A->n_rows;
    std::vector<int> ptr, adj;
    symmetric_adjacency(A, ptr, adj);
    auto degree = [&](int v) { return ptr[v + 1] - ptr[v]; };
    std::vector<char> seen(n, 0);
    perm.clear();
    perm.reserve(n);
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    if (rcm) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return degree(a) < degree(b); });
    std::vector<int> nb;
    for (int start : order) {
        if (seen[start]) continue;
        seen[start] = 1;
        size_t head = perm.size();
        perm.push_back(start);
        while (head < perm.size()) {
            const int v = perm[head++];
            nb.clear();
            for (int k = ptr[v]; k < ptr[v + 1]; ++k) {
                const int w = adj[k];
                if (!seen[w]) { seen[w] = 1; nb.push_back(w); }
            }
            if (rcm) std::stable_sort(nb.begin(), nb.end(), [&](int a, int b) { return degree(a) < degree(b); });
            perm.insert(perm.end(), nb.begin(), nb.end());
        }
    }
    if (rcm) std::reverse(perm.begin(), perm.end());
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
    B->row_ptr = new crs_index[n + 1];
    B->col = new int[A->nnz ? A->nnz : 1];
    B->val = new double[A->nnz ? A->nnz : 1];
    B->row_ptr[0] = 0;
    for (int i = 0; i < n; ++i) B->row_ptr[i + 1] = B->row_ptr[i] + (A->row_ptr[perm[i] + 1] - A->row_ptr[perm[i]]);
    for (int i = 0; i < n; ++i) {
        crs_index p = B->row_ptr[i];
        for (crs_index k = A->row_ptr[perm[i]]; k < A->row_ptr[perm[i] + 1]; ++k) {
            B->col[p] = inv_perm[A->col[k]];
            B->val[p++] = A->val[k];
        }
    }
}

inline void write_permutation(const std::string &path, const std::vector<int> &perm) {
    std::ofstream f(path);
    for (int v : perm) f << v << "\n";
}
