// Host-side nonlinear solver framework with the reference's public interface
// (AbstractNonlinearSolver.hpp:9-42, ConvergenceCriterion.hpp:4-27,
// NewtonSolver.hpp:10-107): damped Newton with an analytic or
// forward-difference Jacobian.  Pure host code on Armadillo types (or the
// stand-in of mi355_arma_compat.hpp); the residual it calls is the GPU path.
#pragma once
#include <string>

#include "nonlinear_problem.hpp"

// ||r||_2 <= tolerance  (ConvergenceCriterion.cpp:11-15)
class ConvergenceCriterion {
  public:
    explicit ConvergenceCriterion(double tolerance) : tol_(tolerance) {}
    bool TestConvergence(double residualNorm) const { return residualNorm <= tol_; }
    void SetTolerance(double tolerance) { tol_ = tolerance; }

  private:
    double tol_;
};

class AbstractNonlinearSolver {
  public:
    enum class ExitFlagType { converged, notConverged };
    virtual ~AbstractNonlinearSolver() {}
    virtual void Solve(arma::vec& solution, arma::vec& residualHistory, ExitFlagType& exitFlag,
                       arma::mat* pJacobianExternal = nullptr) = 0;

  protected:
    // stdout table of AbstractNonlinearSolver.cpp:11-95
    virtual void PrintHeader(const std::string& solverName, int maxIterations, double tolerance) const;
    virtual void PrintFooter(int iteration, ExitFlagType exitFlag) const;
    virtual void PrintIteration(int iteration, double errorEstimate, bool initialise = false) const;
};

class NewtonSolver : public AbstractNonlinearSolver {
  public:
    struct ParameterList {     // defaults of NewtonSolver.hpp:19-26
        double tolerance = 1e-5;
        int maxIterations = 10;
        bool printOutput = true;
        double finiteDifferenceEpsilon = 1e-8;
        double damping = 1.0;
    };

    // The solver keeps NON-OWNING pointers; the parameter list is read live at every Solve()
    // (NewtonSolver.cpp:12-16,222-228), so edits made after construction are honoured.
    NewtonSolver(AbstractNonlinearProblem* pProblem, const arma::vec* pInitialGuess, const ParameterList* pParameterList);
    NewtonSolver(AbstractNonlinearProblem* pProblem, AbstractNonlinearProblemJacobian* pProblemJacobian,
                 const arma::vec* pInitialGuess, const ParameterList* pParameterList);
    ~NewtonSolver();

    void Solve(arma::vec& solution, arma::vec& residualHistory, ExitFlagType& exitFlag,
               arma::mat* pJacobianExternal = nullptr) override;

    void SetInitialGuess(const arma::vec* pInitialGuess) { guess_ = pInitialGuess; }
    void SetParameterList(const ParameterList* pParameterList) { pars_ = pParameterList; }
    void SetProblem(AbstractNonlinearProblem* pProblem) { problem_ = pProblem; }
    void SetProblemJacobian(AbstractNonlinearProblemJacobian* pProblemJacobian) { jacobian_ = pProblemJacobian; }
    void PostProcess();

    int LastIterationCount() const { return last_iterations_; }
    int LastResidualEvaluations() const { return last_evaluations_; }

  private:
    void ForwardDifferenceJacobian(const arma::vec& u, const arma::vec& f, arma::mat& J);

    AbstractNonlinearProblem* problem_;
    AbstractNonlinearProblemJacobian* jacobian_;
    const arma::vec* guess_;
    const ParameterList* pars_;
    ConvergenceCriterion* criterion_;
    int last_iterations_ = 0;
    int last_evaluations_ = 0;
};
