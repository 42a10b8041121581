// sparse_matrix.hpp -- MatrixCRS / MatrixCOO of the MI355X build.
//
// MatrixCRS keeps the reference's layout (sparse_matrix.hpp:59-66: int32
// row_ptr/col, fp64 val) on the host for the setup steps, and a device mirror
// (bis_mat) that every kernel uses.  MatrixCOO::read_from_mtx reproduces the
// reference reader's ordering semantics (sparse_matrix.hpp:261-350): symmetric
// files are expanded with the mirrored entry right after its source entry and
// the triplets are STABLY sorted by row only, so the column order inside a row
// -- and with it the summation order of SpMV / SpTRSV -- is the file order.
#pragma once

#include <algorithm>
#include <cstring>
#include <numeric>
#include <sstream>
#include <stdexcept>

#include "common.hpp"

struct MatrixCRS {
    int n_rows{}, n_cols{}, nnz{};
    int *row_ptr = nullptr;
    int *col = nullptr;
    double *val = nullptr;
    bis_mat *dev = nullptr; // device mirror (owned)

    MatrixCRS() = default;
    MatrixCRS(std::size_t r, std::size_t c, std::size_t z) : n_rows(r), n_cols(c), nnz(z) {
        row_ptr = new int[n_rows + 1];
        col = new int[nnz ? nnz : 1];
        val = new double[nnz ? nnz : 1];
    }
    MatrixCRS(const MatrixCRS &) = delete;
    MatrixCRS &operator=(const MatrixCRS &o) { // deep copy of the host arrays (:92-128)
        if (this == &o) return *this;
        free_host();
        n_rows = o.n_rows; n_cols = o.n_cols; nnz = o.nnz;
        if (o.row_ptr) {
            row_ptr = new int[n_rows + 1];
            col = new int[nnz ? nnz : 1];
            val = new double[nnz ? nnz : 1];
            std::memcpy(row_ptr, o.row_ptr, sizeof(int) * (n_rows + 1));
            std::memcpy(col, o.col, sizeof(int) * nnz);
            std::memcpy(val, o.val, sizeof(double) * nnz);
        }
        return *this;
    }
    // (re)build the device mirror from the host arrays
    void upload() {
        release_device();
        bis::check(bis_mat_create(bis::ctx(), n_rows, n_cols, nnz, row_ptr, col, val, &dev), "bis_mat_create");
    }
    // adopt a matrix that was generated on the device; host arrays stay empty
    void adopt(bis_mat *m) {
        release_device();
        dev = m;
        int64_t r, c, z;
        bis_mat_info(m, &r, &c, &z);
        n_rows = (int)r; n_cols = (int)c; nnz = (int)z;
    }
    void release_device() { if (dev) { bis_mat_destroy(bis::ctx(), dev); dev = nullptr; } }
    void free_host() { delete[] row_ptr; delete[] col; delete[] val; row_ptr = col = nullptr; val = nullptr; }
    ~MatrixCRS() { release_device(); free_host(); }
};

struct MatrixCOO {
    long n_rows{}, n_cols{}, nnz{};
    bool is_sorted{}, is_symmetric{};
    std::vector<int> I, J;
    std::vector<double> values;

    void read_from_mtx(const std::string &path) {
        FILE *f = fopen(path.c_str(), "r");
        if (!f) throw std::runtime_error("Unable to open file: " + path);
        char line[1024];
        if (!fgets(line, sizeof line, f)) { fclose(f); throw std::runtime_error("Could not process Matrix Market banner in file: " + path); }
        std::istringstream hs(line);
        std::string banner, object, format, field, symmetry;
        hs >> banner >> object >> format >> field >> symmetry;
        auto lower = [](std::string &s) { for (auto &c : s) c = (char)tolower(c); };
        lower(object); lower(format); lower(field); lower(symmetry);
        if (banner != "%%MatrixMarket") { fclose(f); throw std::runtime_error("Could not process Matrix Market banner in file: " + path); }
        const bool pattern = field == "pattern";
        const bool numeric = field == "real" || field == "integer";
        const bool symm = symmetry == "symmetric";
        if (object != "matrix" || format != "coordinate" || !(pattern || numeric) || !(symm || symmetry == "general")) {
            fclose(f);
            throw std::runtime_error("Unsupported matrix format in file: " + path);
        }
        long M = 0, N = 0, nz = 0;
        while (fgets(line, sizeof line, f)) {
            if (line[0] == '%') continue;
            if (sscanf(line, "%ld %ld %ld", &M, &N, &nz) == 3) break;
        }
        if (M != N) { fclose(f); throw std::runtime_error("Matrix must be square."); }
        std::vector<int> r, c;
        std::vector<double> v;
        r.reserve(symm ? 2 * nz : nz); c.reserve(r.capacity()); v.reserve(r.capacity());
        for (long e = 0; e < nz; ++e) {
            int i, j;
            double x = 0.01; // pattern entries (mmio.hpp:176-181)
            const int got = pattern ? fscanf(f, "%d %d", &i, &j) : fscanf(f, "%d %d %lg", &i, &j, &x);
            if (got != (pattern ? 2 : 3)) { fclose(f); throw std::runtime_error("Error reading matrix from file: " + path); }
            --i; --j;
            r.push_back(i); c.push_back(j); v.push_back(x);
            if (symm && i != j) { r.push_back(j); c.push_back(i); v.push_back(x); }
        }
        fclose(f);
        std::vector<long> perm(r.size());
        std::iota(perm.begin(), perm.end(), 0L);
        std::stable_sort(perm.begin(), perm.end(), [&](long a, long b) { return r[a] < r[b]; });
        I.resize(r.size()); J.resize(r.size()); values.resize(r.size());
        for (size_t k = 0; k < perm.size(); ++k) { I[k] = r[perm[k]]; J[k] = c[perm[k]]; values[k] = v[perm[k]]; }
        n_rows = M; n_cols = N; nnz = (long)values.size();
        is_sorted = true; is_symmetric = false;
    }
};
