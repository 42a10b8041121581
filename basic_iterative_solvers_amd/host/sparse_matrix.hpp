// sparse_matrix.hpp -- MatrixCRS / MatrixCOO of the MI355X build.
//
// MatrixCRS keeps the reference's layout (sparse_matrix.hpp:59-66: row_ptr, int32
// col, fp64 val; row_ptr and nnz widened to 64 bits) on the host for the setup steps, and a device mirror
// (bis_mat) that every kernel uses.  MatrixCOO::read_from_mtx reproduces the
// reference reader's ordering semantics (sparse_matrix.hpp:261-350): symmetric
// files are expanded with the mirrored entry right after its source entry and
// the triplets are STABLY sorted by row only, so the column order inside a row
// -- and with it the summation order of SpMV / SpTRSV -- is the file order.
#pragma once

#include <algorithm>
#include <charconv>
#include <cstring>
#include <thread>
#include <numeric>
#include <sstream>
#include <stdexcept>

#include "common.hpp"

// nnz and row_ptr are 64-bit here (the reference's are `int`, sparse_matrix.hpp:60-66: it cannot hold HPCG-512's
// 3.6e9 entries); col stays int32 like the reference's and the device layout.
using crs_index = int64_t;

struct MatrixCRS {
    int n_rows{}, n_cols{};
    crs_index nnz{};
    crs_index *row_ptr = nullptr;
    int *col = nullptr;
    double *val = nullptr;
    bis_mat *dev = nullptr; // device mirror (owned)

    MatrixCRS() = default;
    MatrixCRS(std::size_t r, std::size_t c, std::size_t z) : n_rows((int)r), n_cols((int)c), nnz((crs_index)z) {
        row_ptr = new crs_index[n_rows + 1];
        col = new int[nnz ? nnz : 1];
        val = new double[nnz ? nnz : 1];
    }
    MatrixCRS(const MatrixCRS &) = delete;
    MatrixCRS &operator=(const MatrixCRS &o) { // deep copy of the host arrays (:92-128)
        if (this == &o) return *this;
        free_host();
        n_rows = o.n_rows; n_cols = o.n_cols; nnz = o.nnz;
        if (o.row_ptr) {
            row_ptr = new crs_index[n_rows + 1];
            col = new int[nnz ? nnz : 1];
            val = new double[nnz ? nnz : 1];
            std::memcpy(row_ptr, o.row_ptr, sizeof(crs_index) * (n_rows + 1));
            std::memcpy(col, o.col, sizeof(int) * nnz);
            std::memcpy(val, o.val, sizeof(double) * nnz);
        }
        return *this;
    }
    // (re)build the device mirror from the host arrays
    void upload() {
        release_device();
        bis::check(bis_mat_create64(bis::ctx(), n_rows, n_cols, nnz, row_ptr, col, val, &dev), "bis_mat_create");
    }
    // adopt a matrix that was generated on the device; host arrays stay empty
    void adopt(bis_mat *m) {
        release_device();
        dev = m;
        int64_t r, c, z;
        bis_mat_info(m, &r, &c, &z);
        n_rows = (int)r; n_cols = (int)c; nnz = z;
    }
    void release_device() { if (dev) { bis_mat_destroy(bis::ctx(), dev); dev = nullptr; } }
    void free_host() { delete[] row_ptr; delete[] col; delete[] val; row_ptr = nullptr; col = nullptr; val = nullptr; }
    ~MatrixCRS() { release_device(); free_host(); }
};

// MatrixCOO::read_from_mtx: the whole file is read into memory and parsed by a few threads, each on a run of whole
// lines (std::from_chars: no locale, no stdio per entry); the chunks are concatenated in file order, symmetric
// entries are mirrored right after their source entry and the triplets are ordered by row with a stable counting
// sort -- the order the reference's reader produces (stable sort by row only, sparse_matrix.hpp:308-344), so the
// column order inside a row is the file order.  Entry counts are 64-bit.
struct MatrixCOO {
    long n_rows{}, n_cols{}, nnz{};
    bool is_sorted{}, is_symmetric{};
    std::vector<int> I, J;
    std::vector<double> values;

    void read_from_mtx(const std::string &path) {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("Unable to open file: " + path);
        std::string buf;
        {
            fseek(f, 0, SEEK_END);
            const long sz = ftell(f);
            fseek(f, 0, SEEK_SET);
            buf.resize((size_t)std::max(sz, 0L));
            const size_t got = sz > 0 ? fread(&buf[0], 1, (size_t)sz, f) : 0;
            fclose(f);
            if ((long)got != sz) throw std::runtime_error("Error reading matrix from file: " + path);
        }
        const char *p = buf.data(), *end = p + buf.size();
        auto next_line = [&](const char *q) { while (q < end && *q != '\n') ++q; return q < end ? q + 1 : end; };
        // banner
        const char *l0 = p;
        p = next_line(p);
        std::istringstream hs(std::string(l0, p));
        std::string banner, object, format, field, symmetry;
        hs >> banner >> object >> format >> field >> symmetry;
        auto lower = [](std::string &s) { for (auto &c : s) c = (char)tolower(c); };
        lower(object); lower(format); lower(field); lower(symmetry);
        if (banner != "%%MatrixMarket") throw std::runtime_error("Could not process Matrix Market banner in file: " + path);
        const bool pattern = field == "pattern";
        const bool numeric = field == "real" || field == "integer";
        const bool symm = symmetry == "symmetric";
        if (object != "matrix" || format != "coordinate" || !(pattern || numeric) || !(symm || symmetry == "general"))
            throw std::runtime_error("Unsupported matrix format in file: " + path);
        long M = 0, N = 0, nz = 0;
        bool have_size = false;
        while (p < end && !have_size) {
            const char *l = p;
            p = next_line(p);
            if (*l == '%') continue;
            have_size = sscanf(std::string(l, p).c_str(), "%ld %ld %ld", &M, &N, &nz) == 3;
        }
        if (!have_size) throw std::runtime_error("Error reading matrix from file: " + path);
        if (M != N) throw std::runtime_error("Matrix must be square.");
        if (M >= INT32_MAX) throw std::runtime_error("Matrix dimension exceeds the 32-bit column index range: " + path);
        // parse the entries: T threads on runs of whole lines
        const unsigned hw = std::thread::hardware_concurrency();
        const int T = (int)std::max(1u, std::min(16u, std::min(hw ? hw : 1u, (unsigned)((end - p) / (1 << 20)) + 1u)));
        std::vector<const char *> cut(T + 1);
        cut[0] = p; cut[T] = end;
        for (int t = 1; t < T; ++t) cut[t] = next_line(p + (end - p) / T * t - 1);
        struct Part { std::vector<int> r, c; std::vector<double> v; long entries = 0; bool bad = false; };
        std::vector<Part> part(T);
        auto parse = [&](int t) {
            Part &P = part[t];
            const char *q = cut[t], *e = cut[t + 1];
            const size_t guess = (size_t)(e - q) / 12 + 16;
            P.r.reserve(symm ? 2 * guess : guess); P.c.reserve(P.r.capacity()); P.v.reserve(P.r.capacity());
            auto skip = [&]() { while (q < e && (*q == ' ' || *q == '\t' || *q == '\r')) ++q; };
            while (q < e) {
                skip();
                if (q < e && *q == '\n') { ++q; continue; }
                if (q >= e) break;
                if (*q == '%') { q = next_line(q); continue; }
                long i = 0, j = 0;
                double x = 0.01; // pattern entries (mmio.hpp:176-181)
                auto r1 = std::from_chars(q, e, i);
                if (r1.ec != std::errc()) { P.bad = true; return; }
                q = r1.ptr; skip();
                auto r2 = std::from_chars(q, e, j);
                if (r2.ec != std::errc()) { P.bad = true; return; }
                q = r2.ptr;
                if (!pattern) {
                    skip();
                    if (q < e && *q == '+') ++q;
                    auto r3 = std::from_chars(q, e, x);
                    if (r3.ec != std::errc()) { P.bad = true; return; }
                    q = r3.ptr;
                }
                q = next_line(q);
                --i; --j;
                if (i < 0 || i >= M || j < 0 || j >= N) { P.bad = true; return; }
                P.r.push_back((int)i); P.c.push_back((int)j); P.v.push_back(x);
                if (symm && i != j) { P.r.push_back((int)j); P.c.push_back((int)i); P.v.push_back(x); }
                ++P.entries;
            }
        };
        {
            std::vector<std::thread> th;
            for (int t = 1; t < T; ++t) th.emplace_back(parse, t);
            parse(0);
            for (auto &x : th) x.join();
        }
        long entries = 0;
        size_t total = 0;
        for (auto &P : part) {
            if (P.bad) throw std::runtime_error("Error reading matrix from file: " + path);
            entries += P.entries;
            total += P.r.size();
        }
        if (entries != nz) throw std::runtime_error("Error reading matrix from file: " + path);
        buf.clear(); buf.shrink_to_fit();
        // stable counting sort by row over the concatenation (file order inside a row)
        std::vector<int64_t> start((size_t)M + 1, 0);
        for (auto &P : part) for (int r : P.r) ++start[(size_t)r + 1];
        for (long r = 0; r < M; ++r) start[(size_t)r + 1] += start[(size_t)r];
        I.resize(total); J.resize(total); values.resize(total);
        for (auto &P : part) {
            for (size_t k = 0; k < P.r.size(); ++k) {
                const int64_t d = start[(size_t)P.r[k]]++;
                I[(size_t)d] = P.r[k]; J[(size_t)d] = P.c[k]; values[(size_t)d] = P.v[k];
            }
            Part().r.swap(P.r); Part().c.swap(P.c); Part().v.swap(P.v);
        }
        n_rows = M; n_cols = N; nnz = (long)total;
        is_sorted = true; is_symmetric = false;
    }
};
