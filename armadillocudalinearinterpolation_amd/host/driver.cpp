// The reference's fixed problem (Driver.cu:11-126) on the MI355X path: beta = 13.0589, Z0 = (0.3310, 0.6914,
// 1.3557), Newton tolerance 1e-4, max 10 iterations, forward-difference epsilon 1e-2, damping 1, 512 grid points.
//   driver [--real R] [--threads N] [--fast] [--dedup] [--debug DIR] [--json FILE] [--quiet] [--stability]
//          [--reference-mean | --true-mean] [--gpus N | --devices a,b,..]
// --gpus N shards the realisations over GPUs 0..N-1 of the node (mi_group_edm_*: one ComputeF per residual, partial
// sums added across the GPUs); --devices takes explicit ordinals, repeats allowed (0,0 rehearses two shards on one GPU).
// --reference-mean (the default) averages over realisations exactly as EventDrivenMap.cu:800-824 does (realisation 0
// left out of the sum, full count in the divisor); --true-mean sums every accepted realisation.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "event_driven_map.hpp"
#include "newton_solver.hpp"
#include "stability.hpp"

int main(int argc, char* argv[])
{
    unsigned int noReal = 1000;        // Driver.cu:19
    int noThreads = 512;               // Driver.cu:69
    bool fast = false, quiet = false, stability = false, dedup = false, reference_mean = true;
    const char *debug_dir = nullptr, *json = nullptr;
    std::vector<int> devices;          // empty: the reference's single device
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--real") && i + 1 < argc) noReal = std::strtoul(argv[++i], nullptr, 10);
        else if (!std::strcmp(argv[i], "--threads") && i + 1 < argc) noThreads = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--fast")) fast = true;
        else if (!std::strcmp(argv[i], "--dedup")) dedup = true;   // sigma == 0: evolve one realisation, replicate it
        else if (!std::strcmp(argv[i], "--quiet")) quiet = true;
        else if (!std::strcmp(argv[i], "--reference-mean")) reference_mean = true;
        else if (!std::strcmp(argv[i], "--true-mean")) reference_mean = false;
        else if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) { const int n = std::atoi(argv[++i]); devices.clear(); for (int d = 0; d < n; ++d) devices.push_back(d); }
        else if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) {
            devices.clear();
            for (char* tok = std::strtok(argv[++i], ","); tok; tok = std::strtok(nullptr, ",")) devices.push_back(std::atoi(tok));
        }
        else if (!std::strcmp(argv[i], "--stability")) stability = true;
        else if (!std::strcmp(argv[i], "--debug") && i + 1 < argc) debug_dir = argv[++i];
        else if (!std::strcmp(argv[i], "--json") && i + 1 < argc) json = argv[++i];
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }

    arma::vec parameters(1);
    parameters(0) = 13.0589f;                                   // Driver.cu:16 (a float literal)
    EventDrivenMap* p_event = devices.empty() ? new EventDrivenMap(&parameters, noReal)   // Driver.cu:20
                                              : new EventDrivenMap(&parameters, noReal, devices);
    EventDrivenMap& event = *p_event;
    event.SetQuiet(quiet);
    if (fast) event.SetMathMode(MI_EDM_MATH_FAST);
    if (dedup) event.SetDedupIdentical(true);
    event.SetMeanQuirk(reference_mean);

    arma::vec guess(3);                                          // Driver.cu:23-24 (float literals)
    guess(0) = 0.3310f; guess(1) = 0.6914f; guess(2) = 1.3557f;

    NewtonSolver::ParameterList pars;                            // Driver.cu:27-37
    pars.tolerance = 1e-4;
    pars.maxIterations = 10;
    pars.printOutput = !quiet;
    pars.damping = 1.0;
    NewtonSolver newton(&event, &guess, &pars);
    pars.finiteDifferenceEpsilon = 1e-2;                         // set after construction; read live at Solve()

    arma::vec f0(3);
    event.ComputeF(guess, f0);                                   // Driver.cu:59: one residual at N = 1024
    if (!quiet) std::cout << "F(Z0) at 1024 grid points =\n" << f0 << std::endl;

    event.SetNoThreads(noThreads);                               // Driver.cu:69
    if (debug_dir) { event.SetDebugDirectory(debug_dir); event.SetDebugFlag(true); }   // Driver.cu:70
    arma::vec solution(3), history;
    AbstractNonlinearSolver::ExitFlagType flag;
    const auto t0 = std::chrono::steady_clock::now();
    std::string failure;
    flag = AbstractNonlinearSolver::ExitFlagType::notConverged;
    try {
        newton.Solve(solution, history, flag);                   // Driver.cu:71
    } catch (const std::exception& e) {
        // arma::solve throws when the finite-difference Jacobian is singular or NaN (the reference would
        // terminate here); report it as a failed solve instead of aborting
        failure = e.what();
        std::cout << "The method failed: " << failure << std::endl;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!quiet) std::cout << "Homogeneous Solution = \n" << solution << std::endl;

    // Driver.cu:46,86-114 (commented there): unstable eigenvalues of the equation-free map at the solution
    int n_unstable = -1;
    std::vector<std::complex<double>> eigs;
    if (stability && failure.empty()) {
        if (debug_dir) event.SetDebugFlag(false);
        Stability stab(Stability::ProblemType::equationFree, &event);
        stab.SetFiniteDifferenceEpsilon(pars.finiteDifferenceEpsilon);
        eigs = stab.ComputeEigenvalues(solution);
        n_unstable = 0;
        for (const auto& l : eigs) n_unstable += std::abs(l) > 1.0;
        if (!quiet) std::cout << "Number of unstable eigenvalues = " << n_unstable << std::endl;
    }

    const bool ok = flag == AbstractNonlinearSolver::ExitFlagType::converged;
    if (json) {
        FILE* fp = std::fopen(json, "w");
        if (fp) {
            std::fprintf(fp, "{\"converged\": %s, \"iterations\": %d, \"residual_evaluations\": %d, \"n_real\": %u, "
                             "\"n_grid\": %d, \"math\": \"%s\", \"mean\": \"%s\", \"shards\": %d, \"solve_seconds\": %.6f,\n \"solution\": [%.17g, %.17g, %.17g],\n"
                             " \"f0_1024\": [%.17g, %.17g, %.17g],\n \"history\": [",
                         ok ? "true" : "false", newton.LastIterationCount(), newton.LastResidualEvaluations(), noReal,
                         noThreads, fast ? "fast" : "exact", reference_mean ? "reference" : "true", devices.empty() ? 1 : (int)devices.size(), secs, solution(0), solution(1), solution(2), f0(0), f0(1), f0(2));
            const int nh = failure.empty() ? newton.LastIterationCount() + 1 : 0;
            for (int i = 0; i < nh; ++i) std::fprintf(fp, "%s%.17g", i ? ", " : "", history(i));
            std::fprintf(fp, "],\n \"n_unstable\": %d, \"eigenvalues\": [", n_unstable);
            for (size_t i = 0; i < eigs.size(); ++i) std::fprintf(fp, "%s[%.17g, %.17g]", i ? ", " : "", eigs[i].real(), eigs[i].imag());
            std::fprintf(fp, "], \"failure\": \"%s\"}\n", failure.c_str());
            std::fclose(fp);
        }
    }
    delete p_event;
    return ok ? 0 : 1;
}
