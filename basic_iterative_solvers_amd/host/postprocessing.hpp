// postprocessing.hpp -- the stdout contract of the reference
// (postprocessing.hpp:8-68): the residual table `||A*x_k - b||_2 = %.16e` with
// per-iteration seconds, and the summary lines.  This text is the parity
// artefact: histories of the two builds diff textually.
#pragma once

#include "common.hpp"
#include "solver.hpp"

inline void print_residuals(double *norms, double *time_per_iteration, int count, int res_check_len) {
    std::cout << std::scientific << std::setprecision(16);
    std::cout << std::endl;
    std::cout << std::string(15, ' ') << "Residual Norms" << std::string(27, ' ') << "Time for iteration" << std::endl;
    std::cout << "+------------------------------------------+" << std::string(8, ' ')
              << "+-------------------------+" << std::endl;
    for (int i = 0; i < count; ++i) {
        std::cout << "||A*x_" << i * res_check_len << " - b||_2 = " << norms[i];
        if (i > 0) std::cout << std::right << std::setw(30) << time_per_iteration[i + 1] << "[s]";
        std::cout << std::endl;
    }
}

inline void summary_output(Args *, Solver *solver) {
    print_residuals(solver->collected_residual_norms, solver->time_per_iteration,
                    solver->collected_residual_norms_count, solver->residual_check_len);
    if (solver->method == SolverType::GMRES) solver->iter_count += solver->gmres_restart_count;
    std::cout << "\nSolver: " << to_string(solver->method);
    if (solver->method == SolverType::GMRES) std::cout << "(" << solver->gmres_restart_len << ")";
    if (solver->preconditioner != PrecondType::None)
        std::cout << " with preconditioner: " << to_string(solver->preconditioner);
    if (solver->convergence_flag)
        std::cout << " converged in: " << solver->iter_count << " iterations." << std::endl;
    else
        std::cout << " did not converge after " << solver->iter_count << " iterations." << std::endl;
    std::cout << "With the stopping criteria \"tol * ||Ax_0 - b||_2\" is: " << solver->stopping_criteria << std::endl;
    std::cout << "The residual of the final iteration is: ||A*x_star - b||_2 = " << std::scientific
              << solver->collected_residual_norms[solver->collected_residual_norms_count - 1] << ".\n";
}

inline void postprocessing(Args *cli_args, Solver *solver, Timers *) { summary_output(cli_args, solver); }
