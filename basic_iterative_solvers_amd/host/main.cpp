// main.cpp -- CLI driver of the MI355X build; same command line and the same
// stdout (residual table, summary, timer tree) as the reference's main.cpp.
//   ./basic_iterative_solvers <matrix.mtx | generator> <-j|-gs|-sgs|-cg|-gm|-bi>
//        [-p j|gs|bgs|sgs|2st|s2st|ilu0] [-scale 0|1] [-rl N] [-unfused] [-dev K]
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
        if (cli_args->crs_cache.empty() || !read_crs_cache(cli_args->crs_cache, cli_args->matrix_file_name, A.get())) {
            MatrixCOO mtx;
            try {
                mtx.read_from_mtx(cli_args->matrix_file_name);
            } catch (const std::exception &e) {
                fprintf(stderr, "ERROR: %s\n", e.what());
                exit(EXIT_FAILURE);
            }
            convert_coo_to_crs(&mtx, A.get());
            if (!cli_args->crs_cache.empty()) write_crs_cache(cli_args->crs_cache, cli_args->matrix_file_name, A.get());
        }
        A->upload();
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
    }
    bis::shutdown();
    return 0;
}
