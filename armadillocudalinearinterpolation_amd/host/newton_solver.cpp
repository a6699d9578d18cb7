#include "newton_solver.hpp"

#include <cassert>
#include <cmath>
#include <iomanip>
#include <iostream>
#include <limits>

namespace {
const char* const kRule = "------------------------------------------------";
}

void AbstractNonlinearSolver::PrintHeader(const std::string& solverName, int maxIterations, double tolerance) const
{
    std::cout << kRule << '\n'
              << " Attempt to solve nonlinear problem with " << solverName << '\n'
              << " max number of iterations = " << maxIterations << '\n'
              << " tolerance = " << tolerance << '\n'
              << kRule << std::endl;
}

void AbstractNonlinearSolver::PrintFooter(int iteration, ExitFlagType exitFlag) const
{
    std::cout << kRule << '\n'
              << (exitFlag == ExitFlagType::converged ? "The method converged after " : "The method failed to converge after ")
              << iteration << " iterations" << std::endl;
}

void AbstractNonlinearSolver::PrintIteration(int iteration, double errorEstimate, bool initialise) const
{
    if (initialise) std::cout << std::setw(10) << "Iteration" << std::setw(25) << "error estimate" << std::endl;
    std::cout << std::setw(10) << iteration << std::scientific << std::setprecision(6) << std::setw(25) << errorEstimate
              << std::endl;
}

NewtonSolver::NewtonSolver(AbstractNonlinearProblem* pProblem, const arma::vec* pInitialGuess,
                           const ParameterList* pParameterList)
    : problem_(pProblem), jacobian_(nullptr), guess_(pInitialGuess), pars_(pParameterList), criterion_(nullptr)
{
}

NewtonSolver::NewtonSolver(AbstractNonlinearProblem* pProblem, AbstractNonlinearProblemJacobian* pProblemJacobian,
                           const arma::vec* pInitialGuess, const ParameterList* pParameterList)
    : problem_(pProblem), jacobian_(pProblemJacobian), guess_(pInitialGuess), pars_(pParameterList), criterion_(nullptr)
{
}

NewtonSolver::~NewtonSolver() { delete criterion_; }

void NewtonSolver::PostProcess() { problem_->PostProcess(); }

// NewtonSolver.cpp:164-197: column i = (F(u + eps e_i) - F(u)) * eps^-1.
void NewtonSolver::ForwardDifferenceJacobian(const arma::vec& u, const arma::vec& f, arma::mat& J)
{
    const arma::uword n = guess_->n_rows;
    const double eps = pars_->finiteDifferenceEpsilon;
    const double inv_eps = std::pow(eps, -1);
    if (auto* batched = dynamic_cast<AbstractBatchedNonlinearProblem*>(problem_)) {
        // the n perturbed residuals are independent: hand them over together (same values as the loop below)
        arma::mat U(n, n), F;
        for (arma::uword i = 0; i < n; ++i) {
            for (arma::uword r = 0; r < n; ++r) U(r, i) = u(r);
            U(i, i) += eps;
        }
        batched->ComputeFBatch(U, F);
        last_evaluations_ += (int)n;
        for (arma::uword i = 0; i < n; ++i)
            for (arma::uword r = 0; r < n; ++r) J(r, i) = (F(r, i) - f(r)) * inv_eps;
        return;
    }
    arma::vec du(u), df(n);
    for (arma::uword i = 0; i < n; ++i) {
        if (i > 0) du(i - 1) = u(i - 1);     // undo the previous column's perturbation
        du(i) += eps;
        problem_->ComputeF(du, df);
        ++last_evaluations_;
        for (arma::uword r = 0; r < n; ++r) J(r, i) = (df(r) - f(r)) * inv_eps;
    }
}

// NewtonSolver.cpp:40-161
void NewtonSolver::Solve(arma::vec& solution, arma::vec& residualHistory, ExitFlagType& exitFlag,
                         arma::mat* pJacobianExternal)
{
    const int max_it = pars_->maxIterations;
    const bool print = pars_->printOutput;
    const double tol = pars_->tolerance;
    if (criterion_) criterion_->SetTolerance(tol);
    else criterion_ = new ConvergenceCriterion(tol);
    if (print) PrintHeader("Newton Method", max_it, tol);

    const arma::uword n = guess_->n_rows;
    assert(n == solution.n_rows);
    last_evaluations_ = 0;
    int it = 0;
    solution = *guess_;
    arma::vec residual(n);
    problem_->ComputeF(solution, residual);
    ++last_evaluations_;
    double rnorm = arma::norm(residual, 2);

    // The reference sizes the history 1 + maxIterations and never trims it (NewtonSolver.cpp:73,134:
    // the result of head() is discarded); entries past the last iteration are NaN here, garbage there.
    residualHistory.set_size(1 + max_it);
    residualHistory.fill(std::numeric_limits<double>::quiet_NaN());
    residualHistory(0) = rnorm;
    if (print) PrintIteration(it, rnorm, true);
    bool converged = criterion_->TestConvergence(rnorm);

    arma::mat J(n, n);
    J.zeros();
    arma::vec rhs(n);
    while (it < max_it && !converged) {
        if (jacobian_) jacobian_->ComputeDFDU(solution, J);
        else ForwardDifferenceJacobian(solution, residual, J);
        for (arma::uword r = 0; r < n; ++r) rhs(r) = -residual(r);
        const arma::vec direction = arma::solve(J, rhs);
        for (arma::uword r = 0; r < n; ++r) solution(r) += pars_->damping * direction(r);
        ++it;
        problem_->ComputeF(solution, residual);
        ++last_evaluations_;
        rnorm = arma::norm(residual, 2);
        converged = criterion_->TestConvergence(rnorm);
        residualHistory(it) = rnorm;
        if (print) PrintIteration(it, rnorm);
    }
    PostProcess();
    last_iterations_ = it;
    exitFlag = converged ? ExitFlagType::converged : ExitFlagType::notConverged;
    if (print) PrintFooter(it, exitFlag);
    if (pJacobianExternal) {
        assert(pJacobianExternal->n_rows == n && pJacobianExternal->n_cols == n);
        *pJacobianExternal = J;
    }
}
