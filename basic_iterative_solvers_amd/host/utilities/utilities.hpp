// utilities.hpp -- CLI, timer-tree printing and COO->CRS conversion
// (reference utilities/utilities.hpp:12-108, :154-324, :326-367).  Same flags
// and the same output layout.  Extra, MI355X-specific: the <matrix> argument
// may be a generator string instead of a file (the reference does the same
// with SCAMAC strings when built with -DUSE_SCAMAC, main.cpp:48-54):
//     hpcg:N | hpcg:NX,NY,NZ | anderson:L[,shift=S][,W=w][,t=t][,seed=k] | fem:NX[,NY,NZ][,keep=K][,seed=k] | unstr:NX[,NY,NZ][,keep=K][,seed=k]
// and `-unfused` / `-dev K` select the kernel-by-kernel CG and the device;
// `-perm mc` applies the multi-colour reordering of utilities/permute.hpp.
#pragma once

#include <sys/stat.h>

#include "../common.hpp"
#include "../sparse_matrix.hpp"

inline void parse_cli(Args *a, int argc, char *argv[]) {
    if (argc < 3) {
        printf("ERROR: parse_cli: Not enough arguments given. A call should contain:"
               "\n%s <matrix> <method> [extra_features]\n", argv[0]);
        exit(EXIT_FAILURE);
    }
    a->matrix_file_name = argv[1];
    const std::string st = argv[2];
    if (st == "-j") a->method = SolverType::Jacobi;
    else if (st == "-gs") a->method = SolverType::GaussSeidel;
    else if (st == "-sgs") a->method = SolverType::SymmetricGaussSeidel;
    else if (st == "-cg") a->method = SolverType::ConjugateGradient;
    else if (st == "-gm") a->method = SolverType::GMRES;
    else if (st == "-bi") a->method = SolverType::BiCGSTAB;
    else {
        printf("ERROR: parse_cli: Please choose an available solver:"
               "\n-j (Jacobi)\n-gs (Gauss-Seidel)\n-sgs (Symmetric Gauss-Seidel)"
               "\n-gm ([Preconditioned] GMRES)\n-cg ([Preconditioned] Conjugate Gradient)"
               "\n-bi ([Preconditioned] BiCGSTAB)\n");
        exit(EXIT_FAILURE);
    }
    for (int i = 3; i < argc; ++i) {
        const std::string arg = argv[i];
        if (arg == "-p") {
            if (i + 1 >= argc) {
                printf("ERROR: parse_cli: Not enough arguments given. Some extra features need "
                       "additional arguments. Example:\n%s <matrix> <method> -p gs\n", argv[0]);
                exit(EXIT_FAILURE);
            }
            const std::string pt = argv[++i];
            static const std::map<std::string, PrecondType> pcs = {
                {"j", PrecondType::Jacobi}, {"gs", PrecondType::GaussSeidel},
                {"bgs", PrecondType::BackwardsGaussSeidel}, {"sgs", PrecondType::SymmetricGaussSeidel},
                {"2st", PrecondType::TwoStageGS}, {"s2st", PrecondType::SymmetricTwoStageGS},
                {"ilu0", PrecondType::ILU0}};
            auto it = pcs.find(pt);
            if (it == pcs.end()) {
                fprintf(stderr, "ERROR: assign_cli_inputs: Please choose an available preconditioner type: "
                                "\n-p j (Jacobi)\n-p gs (Gauss-Seidel)\n-p bgs (Backwards Gauss-Seidel)"
                                "\n-p sgs (Symmetric Gauss-Seidel)\n-p 2st (2 Stage Gauss-Seidel)"
                                "\n-p s2st (Symmetric 2 Stage Gauss-Seidel)\n-p ilu0 (Incomplete LU with 0 fill-in)\n");
                exit(EXIT_FAILURE);
            }
            a->preconditioner = it->second;
        } else if (arg == "-scale" && i + 1 < argc) a->num_scale = (bool)atoi(argv[++i]);
        else if (arg == "-rl" && i + 1 < argc) a->restart_length = atoi(argv[++i]);
        else if (arg == "-unfused") a->unfused = true;
        else if (arg == "-hostscalars") a->host_scalars = true;
        else if (arg == "-trsv" && i + 1 < argc) {
            const std::string m = argv[++i];
            a->trsv_mode = m == "tiled" ? 1 : m == "level" ? 0 : m == "chain" ? 2 : m == "wave" ? 3 : -1;
        } else if (arg == "-grid" && i + 1 < argc) {
            a->grid_hint[3] = 1;
            if (sscanf(argv[++i], "%lld,%lld,%lld,%lld", &a->grid_hint[0], &a->grid_hint[1], &a->grid_hint[2], &a->grid_hint[3]) < 3) {
                fprintf(stderr, "ERROR: -grid NX,NY,NZ[,DOF]\n");
                exit(EXIT_FAILURE);
            }
        }
        else if (arg == "-perm" && i + 1 < argc) a->perm_mode = argv[++i];
        else if (arg == "-dump-perm" && i + 1 < argc) a->dump_perm = argv[++i];
        else if (arg == "-dump-x" && i + 1 < argc) a->dump_x = argv[++i];
        else if (arg == "-perm-host") a->perm_host = true;
        else if (arg == "-cache" && i + 1 < argc) a->crs_cache = argv[++i];
        else if (arg == "-dev" && i + 1 < argc) a->device = atoi(argv[++i]);
        else std::cout << "ERROR: assign_cli_inputs: Arguement \"" << arg << "\" not recongnized." << std::endl;
    }
}

inline void print_timers(Args *cli_args, Timers *timers) {
    auto line = [&](const char *label, const char *key) {
        std::cout << std::left << std::setw(25) << label << std::right << std::setw(30)
                  << (*timers)[key].get_wtime() << "[s]" << std::endl;
    };
    const SolverType m = cli_args->method;
    std::cout << std::endl << std::scientific << std::setprecision(3);
    std::cout << "+---------------------------------------------------------+" << std::endl;
    line("Total elapsed time: ", "total");
    line("| Preprocessing time: ", "preprocessing");
    line("| | Init time: ", "preprocessing_init");
    line("| | Factor time: ", "preprocessing_factor");
    line("| Solve time: ", "solve");
    line("| | Iterate time: ", "iterate");
    line("| | | SpMV time: ", "spmv");
    line("| | | Precond. time: ", "precond");
    if (m == SolverType::Jacobi) line("| | | Normalize time: ", "normalize");
    else if (m == SolverType::GaussSeidel || m == SolverType::SymmetricGaussSeidel) {
        line("| | | Sum time: ", "sum");
        line("| | | SpTRSV time: ", "sptrsv");
    } else if (m == SolverType::ConjugateGradient || m == SolverType::BiCGSTAB) {
        line("| | | Dot time: ", "dot");
        line("| | | Sum time: ", "sum");
    } else if (m == SolverType::GMRES) {
        line("| | | Orthog. time: ", "orthog");
        line("| | | | Dot time: ", "dot");
        line("| | | | Sum time: ", "sum");
        line("| | | | Norm time: ", "norm");
        line("| | | | Scale time: ", "scale");
        line("| | | Least Sq. time: ", "least_sq");
        line("| | | | DGEMM time: ", "dgemm");
        line("| | | Update g time: ", "update_g");
        line("| | | | DGEMV time: ", "dgemv");
    }
    line("| | Sample time: ", "sample");
    line("| | Exchange time: ", "exchange");
    if (m == SolverType::GMRES) line("| | Restart time: ", "restart");
    line("| | Save x* time: ", "save_x_star");
    line("| Postprocessing time: ", "postprocessing");
    std::cout << "+---------------------------------------------------------+" << std::endl << std::endl;
}

inline void convert_coo_to_crs(MatrixCOO *coo, MatrixCRS *crs) {
    crs->n_rows = (int)coo->n_rows; crs->n_cols = (int)coo->n_cols; crs->nnz = coo->nnz;
    crs->row_ptr = new crs_index[crs->n_rows + 1]();
    crs->col = new int[crs->nnz ? crs->nnz : 1];
    crs->val = new double[crs->nnz ? crs->nnz : 1];
    for (crs_index k = 0; k < crs->nnz; ++k) { crs->col[k] = coo->J[k]; crs->val[k] = coo->values[k]; ++crs->row_ptr[coo->I[k] + 1]; }
    for (int r = 0; r < crs->n_rows; ++r) crs->row_ptr[r + 1] += crs->row_ptr[r];
    if (crs->row_ptr[crs->n_rows] != crs->nnz) { printf("ERROR: converting to CRS.\n"); exit(1); }
}

// Binary CRS cache of a parsed .mtx input: header {magic, n_rows, n_cols, nnz, size and
// modification time of the .mtx it was made from} as int64, then row_ptr (int64[n_rows+1]), col (int32[nnz]),
// val (double[nnz]) exactly as convert_coo_to_crs produced them, so a cached run sees the same matrix bit for bit.
// A cache that does not belong to the named .mtx (other size / time stamp), is truncated, or whose structure is
// not a CRS (row_ptr not monotone from 0 to nnz, column out of range) is ignored and rewritten.
inline bool source_stamp(const std::string &mtx, long long &size, long long &mtime) {
    struct stat st;
    if (stat(mtx.c_str(), &st) != 0) return false;
    size = (long long)st.st_size; mtime = (long long)st.st_mtime;
    return true;
}
inline bool read_crs_cache(const std::string &path, const std::string &mtx, MatrixCRS *A) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    long long h[6] = {0, 0, 0, 0, 0, 0}, sz = 0, mt = 0;
    bool ok = fread(h, sizeof(long long), 6, f) == 6 && h[0] == 0x4253435232ll && source_stamp(mtx, sz, mt) && h[4] == sz && h[5] == mt &&
              h[1] >= 0 && h[1] < INT32_MAX && h[2] == h[1] && h[3] >= 0;
    if (ok) { // the payload must be exactly what the header promises
        const long long want = 48 + 8 * (h[1] + 1) + 12 * h[3];
        struct stat st;
        ok = stat(path.c_str(), &st) == 0 && (long long)st.st_size == want;
    }
    if (ok) {
        MatrixCRS tmp((std::size_t)h[1], (std::size_t)h[2], (std::size_t)h[3]);
        ok = fread(tmp.row_ptr, sizeof(crs_index), (size_t)h[1] + 1, f) == (size_t)h[1] + 1 &&
             fread(tmp.col, sizeof(int), (size_t)h[3], f) == (size_t)h[3] &&
             fread(tmp.val, sizeof(double), (size_t)h[3], f) == (size_t)h[3];
        ok = ok && tmp.row_ptr[0] == 0 && tmp.row_ptr[h[1]] == h[3];
        for (long long r = 0; ok && r < h[1]; ++r) ok = tmp.row_ptr[r] <= tmp.row_ptr[r + 1];
        for (long long k = 0; ok && k < h[3]; ++k) ok = tmp.col[k] >= 0 && tmp.col[k] < h[2];
        if (ok) *A = tmp;
    }
    fclose(f);
    return ok;
}
inline void write_crs_cache(const std::string &path, const std::string &mtx, const MatrixCRS *A) {
    long long sz = 0, mt = 0;
    if (!source_stamp(mtx, sz, mt)) return;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "WARNING: cannot write the CRS cache %s\n", path.c_str()); return; }
    const long long h[6] = {0x4253435232ll, A->n_rows, A->n_cols, A->nnz, sz, mt};
    fwrite(h, sizeof(long long), 6, f);
    fwrite(A->row_ptr, sizeof(crs_index), (size_t)A->n_rows + 1, f);
    fwrite(A->col, sizeof(int), (size_t)A->nnz, f);
    fwrite(A->val, sizeof(double), (size_t)A->nnz, f);
    fclose(f);
}

// generator strings -> device-resident matrix (no host copy)
inline bool make_generated_matrix(const std::string &spec, MatrixCRS *A) {
    auto split = [](const std::string &s, char d) {
        std::vector<std::string> out; std::stringstream ss(s); std::string t;
        while (std::getline(ss, t, d)) out.push_back(t);
        return out;
    };
    const auto head = split(spec, ':');
    if (head.size() != 2) return false;
    bis_mat *m = nullptr;
    const auto f = split(head[1], ',');
    if (head[0] == "hpcg") {
        long nx = atol(f[0].c_str()), ny = nx, nz = nx;
        if (f.size() == 3) { ny = atol(f[1].c_str()); nz = atol(f[2].c_str()); }
        bis::check(bis_mat_gen_hpcg(bis::ctx(), nx, ny, nz, 0, nx * ny * nz, &m), "bis_mat_gen_hpcg");
    } else if (head[0] == "anderson") {
        const long L = atol(f[0].c_str());
        double t = 1.0, W = 5.0, shift = 0.0;
        unsigned long long seed = 1;
        for (size_t i = 1; i < f.size(); ++i) {
            const auto kv = split(f[i], '=');
            if (kv.size() != 2) continue;
            if (kv[0] == "shift") shift = atof(kv[1].c_str());
            else if (kv[0] == "W") W = atof(kv[1].c_str());
            else if (kv[0] == "t") t = atof(kv[1].c_str());
            else if (kv[0] == "seed") seed = strtoull(kv[1].c_str(), nullptr, 10);
        }
        bis::check(bis_mat_gen_anderson(bis::ctx(), L, t, W, shift, seed, 0, L * L * L, &m), "bis_mat_gen_anderson");
    } else if (head[0] == "fem" || head[0] == "unstr") { // fem:NX[,NY,NZ][,keep=K][,seed=S]  (3 unknowns per node); unstr: the same
                                                          // under a seeded random row permutation, no grid (bis_mat_gen_unstr)
        long nx = atol(f[0].c_str()), ny = nx, nz = nx;
        int keep = 85;
        unsigned long long seed = 1;
        size_t i = 1;
        if (f.size() >= 3 && f[1].find('=') == std::string::npos && f[2].find('=') == std::string::npos) {
            ny = atol(f[1].c_str()); nz = atol(f[2].c_str()); i = 3;
        }
        for (; i < f.size(); ++i) {
            const auto kv = split(f[i], '=');
            if (kv.size() != 2) continue;
            if (kv[0] == "keep") keep = atoi(kv[1].c_str());
            else if (kv[0] == "seed") seed = strtoull(kv[1].c_str(), nullptr, 10);
        }
        if (head[0] == "unstr") bis::check(bis_mat_gen_unstr(bis::ctx(), nx, ny, nz, keep, seed, nullptr, &m), "bis_mat_gen_unstr");
        else bis::check(bis_mat_gen_fem(bis::ctx(), nx, ny, nz, keep, seed, 0, 3 * nx * ny * nz, &m), "bis_mat_gen_fem");
    } else return false;
    A->adopt(m);
    return true;
}
