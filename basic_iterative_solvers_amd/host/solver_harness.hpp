// solver_harness.hpp -- the main loop, reference solver_harness.hpp:7-61:
// iterate, sample the residual, milestone prints (res3 / res6 with the timer
// tree), exchange, restart check, until the stopping test fires; then record
// the final solution and its true residual.
#pragma once

#include "common.hpp"
#include "solver.hpp"
#include "utilities/utilities.hpp"

inline void solve(Args *cli_args, Solver *solver, Timers *timers) {
    const double initial_res = solver->collected_residual_norms[0];
    bool res_3 = false, res_6 = false;
    do {
        timers->per_iteration_time->start();
        TIME(timers, "iterate", solver->iterate(timers))
        ++solver->iter_count;
        TIME(timers, "sample", solver->sample_residual(timers->per_iteration_time))
        if (solver->residual_norm / initial_res < 1e-3 && !res_3) {
            std::cout << "res3 => iter_count: " << solver->iter_count << std::endl;
            print_timers(cli_args, timers);
            res_3 = true;
        }
        if (solver->residual_norm / initial_res < 1e-6 && !res_6) {
            std::cout << "res6 => iter_count: " << solver->iter_count << std::endl;
            print_timers(cli_args, timers);
            res_6 = true;
        }
        TIME(timers, "exchange", solver->exchange())
        TIME(timers, "restart", solver->check_restart(timers))
    } while (!solver->check_stopping_criteria());
    if (solver->residual_norm < solver->stopping_criteria) solver->convergence_flag = true;
    TIME(timers, "save_x_star", solver->save_x_star())
    // -perm: the solve ran on P A P^T; hand x* back in the caller's row order (the reference's SMAX path
    // leaves it permuted, smax_helpers.hpp:44-80)
    solver->unpermute_x_star();
    if (!cli_args->dump_x.empty()) { // -dump-x FILE: x* as text, one %.17g value per line (tests)
        std::vector<double> xs(solver->N);
        to_host(xs.data(), solver->x_star, solver->N);
        FILE *f = fopen(cli_args->dump_x.c_str(), "w");
        if (!f) { fprintf(stderr, "ERROR: cannot write %s\n", cli_args->dump_x.c_str()); exit(EXIT_FAILURE); }
        for (double v : xs) fprintf(f, "%.17g\n", v);
        fclose(f);
    }
}
