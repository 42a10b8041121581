// main.cpp -- CLI driver of the MI355X build; same command line and the same
// stdout (residual table, summary, timer tree) as the reference's main.cpp.
//   ./basic_iterative_solvers <matrix.mtx | generator> <-j|-gs|-sgs|-cg|-gm|-bi>
//        [-p j|gs|bgs|sgs|2st|s2st|ilu0] [-scale 0|1] [-rl N] [-unfused] [-dev K]
#include <chrono>

#include "common.hpp"
#include "methods/bicgstab.hpp"
#include "methods/cg.hpp"
#include "methods/gauss_seidel.hpp"
#include "methods/gmres.hpp"
#include "methods/jacobi.hpp"
#include "postprocessing.hpp"
#include "preprocessing.hpp"
#include "solver_harness.hpp"
#include "utilities/utilities.hpp"

static std::string g_input_line; // how long the input phase took (file inputs only)

static void run(Args *cli_args, Timers *timers) {
    Solver *solver = nullptr;
    switch (cli_args->method) {
    case SolverType::Jacobi: solver = new JacobiSolver(cli_args); break;
    case SolverType::GaussSeidel: solver = new GaussSeidelSolver(cli_args); break;
    case SolverType::SymmetricGaussSeidel: solver = new SymmetricGaussSeidelSolver(cli_args); break;
    case SolverType::ConjugateGradient: solver = new ConjugateGradientSolver(cli_args); break;
    case SolverType::GMRES: solver = new GMRESSolver(cli_args); break;
    case SolverType::BiCGSTAB: solver = new BiCGSTABSolver(cli_args); break;
    }
    auto A = std::make_unique<MatrixCRS>();
    if (!make_generated_matrix(cli_args->matrix_file_name, A.get())) {
        // the input phase (sparse_matrix.hpp:225-357, utilities.hpp:326-367 of the reference), timed step by step: g_input_line is
        // printed behind the reference's output
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t0 = now(), t_read = 0.0, t_conv = 0.0, t_cache_w = 0.0, t_cache_r = 0.0;
        bool from_cache = false;
        if (!cli_args->crs_cache.empty()) {
            from_cache = read_crs_cache(cli_args->crs_cache, cli_args->matrix_file_name, A.get());
            t_cache_r = now() - t0;
        }
        if (!from_cache) {
            MatrixCOO mtx;
            t0 = now();
            try {
                mtx.read_from_mtx(cli_args->matrix_file_name);
            } catch (const std::exception &e) {
                fprintf(stderr, "ERROR: %s\n", e.what());
                exit(EXIT_FAILURE);
            }
            t_read = now() - t0; t0 = now();
            convert_coo_to_crs(&mtx, A.get());
            t_conv = now() - t0; t0 = now();
            if (!cli_args->crs_cache.empty()) write_crs_cache(cli_args->crs_cache, cli_args->matrix_file_name, A.get());
            t_cache_w = now() - t0;
        }
        t0 = now();
        A->upload();
        bis_sync(bis::ctx());
        const double t_up = now() - t0;
        char line[512];
        if (from_cache) snprintf(line, sizeof line, "Matrix input: binary CRS cache read %.3f s, upload %.3f s (%lld rows, %lld non-zeros)", t_cache_r, t_up, (long long)A->n_rows, (long long)A->nnz);
        else snprintf(line, sizeof line, "Matrix input: .mtx read + parse %.3f s, COO -> CRS %.3f s, cache write %.3f s, upload %.3f s (%lld rows, %lld non-zeros)",
                      t_read, t_conv, t_cache_w, t_up, (long long)A->n_rows, (long long)A->nnz);
        g_input_line = line;
    }
    if (cli_args->grid_hint[0] > 0)
        bis::check(bis_mat_set_grid_hint(A->dev, cli_args->grid_hint[0], cli_args->grid_hint[1], cli_args->grid_hint[2], (int)cli_args->grid_hint[3]),
                   "-grid (the extents must multiply to the number of rows)");
    TIME(timers, "preprocessing", preprocessing(cli_args, solver, timers, A))
    TIME(timers, "solve", solve(cli_args, solver, timers))
    TIME(timers, "postprocessing", postprocessing(cli_args, solver, timers))
    delete solver;
}

int main(int argc, char *argv[]) {
    Timers timers;
    Args cli_args;
    parse_cli(&cli_args, argc, argv);
    bis::init(cli_args.device);
    // -trsv tiled | level (no tiled sweep: chained where its plan applies, else level-scheduled) | chain (no tiled sweep, chained also with
    // short chains) | wave (neither: the level-scheduled kernels of round 1)
    if (cli_args.trsv_mode == 0 || cli_args.trsv_mode == 1) bis_set_option("trsv_tiled", cli_args.trsv_mode);
    if (cli_args.trsv_mode == 2) { bis_set_option("trsv_tiled", 0); bis_set_option("trsv_chain", 1); }
    if (cli_args.trsv_mode == 3) { bis_set_option("trsv_tiled", 0); bis_set_option("trsv_chain", 0); }
    TIME(&timers, "total", run(&cli_args, &timers))
    print_timers(&cli_args, &timers);
    {   // one line BEHIND the reference's output: the library options in effect (bis_set_option state and the BIS_* variables
        // found in the environment), so that a record of this run says which kernels it selected
        char opts[2048];
        bis_options_describe(opts, (int)sizeof opts);
        std::cout << "Device library options in effect: " << opts << std::endl;
        if (!g_input_line.empty()) std::cout << g_input_line << std::endl;
    }
    bis::shutdown();
    return 0;
}
