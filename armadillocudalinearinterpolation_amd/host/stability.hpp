// Stability: number of unstable eigenvalues of the linearisation at a solution, with the reference's
// interface (Stability.hpp:8-52, Stability.cpp:22-111).  `flow`: Re(lambda) > 0; `map` / `equationFree`:
// |lambda| > 1, where equationFree adds the identity to the residual Jacobian first (Stability.cpp:67-70).
// The reference calls arma::eig_gen (LAPACK dgeev); LAPACK is not available in this image, so eigenvalues
// come from the small complex shifted-QR routine below (mi355::eig_general), used with or without Armadillo.
#pragma once
#include <complex>
#include <vector>

#include "nonlinear_problem.hpp"

namespace mi355 {
// eigenvalues of a real n x n matrix given column-major (arma::mat memptr order); throws if QR stalls
std::vector<std::complex<double>> eig_general(const double* a_colmajor, int n);
}

class Stability {
  public:
    enum class ProblemType { flow, map, equationFree };

    Stability(ProblemType type, AbstractNonlinearProblem* pProblem);
    Stability(ProblemType type, AbstractNonlinearProblem* pProblem, AbstractNonlinearProblemJacobian* pProblemJacobian);

    int ComputeNumUnstableEigenvalues(const arma::vec& u);
    int ComputeNumUnstableEigenvalues(const arma::mat& jacobian);

    // The reference never initialises its finite-difference step (Stability.hpp:50, Stability.cpp:90);
    // here it defaults to NewtonSolver::ParameterList's 1e-8 and can be set.
    void SetFiniteDifferenceEpsilon(double eps) { eps_ = eps; }
    std::vector<std::complex<double>> ComputeEigenvalues(const arma::vec& u);

  private:
    void ForwardDifferenceJacobian(const arma::vec& u, arma::mat& J);
    int Count(const std::vector<std::complex<double>>& ev) const;
    AbstractNonlinearProblem* problem_;
    AbstractNonlinearProblemJacobian* jacobian_;
    ProblemType type_;
    double eps_ = 1e-8;
};
