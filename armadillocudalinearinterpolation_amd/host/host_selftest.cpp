// CPU-only self-test of the host layer (no GPU, no libmi355interp calls): the Armadillo stand-in's
// solve/norm/conv_to and the Newton loop on analytic problems.  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <complex>
#include <vector>

#include "newton_solver.hpp"
#include "stability.hpp"

static int failures = 0;
#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

// F(u) = (u0^2 + u1^2 - 4, u0*u1 - 1, u2 - exp(-u0)) ; counts evaluations
struct Toy : AbstractNonlinearProblem {
    int evals = 0, post = 0;
    void ComputeF(const arma::vec& u, arma::vec& f) override
    {
        ++evals;
        f.set_size(3);
        f(0) = u(0) * u(0) + u(1) * u(1) - 4.0;
        f(1) = u(0) * u(1) - 1.0;
        f(2) = u(2) - std::exp(-u(0));
    }
    void PostProcess() override { ++post; }
};
struct ToyJac : AbstractNonlinearProblemJacobian {
    void ComputeDFDU(const arma::vec& u, arma::mat& J) override
    {
        J.zeros();
        J(0, 0) = 2 * u(0); J(0, 1) = 2 * u(1);
        J(1, 0) = u(1);     J(1, 1) = u(0);
        J(2, 0) = std::exp(-u(0)); J(2, 2) = 1.0;
    }
};

int main()
{
    {   // stand-in / Armadillo basics
        arma::vec v(3);
        v(0) = 3; v(1) = -4; v(2) = 12;
        EXPECT(std::fabs(arma::norm(v, 2) - 13.0) < 1e-14);
        arma::mat A(3, 3);
        const double a[9] = {0, 2, 1, 1, 1, 0, 2, -1, 3};      // column-major; needs a pivot (A(0,0) = 0)
        for (int i = 0; i < 9; ++i) A.memptr()[i] = a[i];
        arma::vec x0(3);
        x0(0) = 1.5; x0(1) = -2; x0(2) = 0.25;
        arma::vec b(3);
        for (int r = 0; r < 3; ++r) { b(r) = 0; for (int c = 0; c < 3; ++c) b(r) += A(r, c) * x0(c); }
        arma::vec x = arma::solve(A, b);
        for (int r = 0; r < 3; ++r) EXPECT(std::fabs(x(r) - x0(r)) < 1e-13);
        arma::fvec fv = arma::conv_to<arma::fvec>::from(v);
        EXPECT(fv(2) == 12.0f && fv.n_elem == 3);
    }
    NewtonSolver::ParameterList pars;
    EXPECT(pars.tolerance == 1e-5 && pars.maxIterations == 10 && pars.finiteDifferenceEpsilon == 1e-8 && pars.damping == 1.0);
    pars.printOutput = false;
    arma::vec guess(3);
    guess(0) = 2.0; guess(1) = 0.4; guess(2) = 0.0;
    const double r0 = std::sqrt(2 + std::sqrt(3.0)), r1 = 1 / r0;   // u0^2+u1^2=4, u0 u1=1
    {   // finite-difference Jacobian path: 1 + it*(n+1) residual evaluations, PostProcess once
        Toy toy;
        NewtonSolver s(&toy, &guess, &pars);
        arma::vec sol(3), hist;
        AbstractNonlinearSolver::ExitFlagType flag;
        s.Solve(sol, hist, flag);
        EXPECT(flag == AbstractNonlinearSolver::ExitFlagType::converged);
        EXPECT(std::fabs(sol(0) - r0) < 1e-7 && std::fabs(sol(1) - r1) < 1e-7 && std::fabs(sol(2) - std::exp(-r0)) < 1e-7);
        EXPECT(toy.evals == 1 + s.LastIterationCount() * 4 && toy.post == 1);
        EXPECT(hist.n_elem == 11);                                   // 1 + maxIterations, untrimmed
        EXPECT(hist(s.LastIterationCount()) <= 1e-5 && hist(0) > 0.1);
        for (int i = 1; i <= s.LastIterationCount(); ++i) EXPECT(hist(i) < hist(i - 1));
    }
    {   // analytic Jacobian path + external Jacobian copy + live parameter list
        Toy toy;
        ToyJac jac;
        NewtonSolver s(&toy, &jac, &guess, &pars);
        pars.tolerance = 1e-12;                                       // edited AFTER construction: must be honoured
        arma::vec sol(3), hist;
        arma::mat Jout(3, 3);
        AbstractNonlinearSolver::ExitFlagType flag;
        s.Solve(sol, hist, flag, &Jout);
        EXPECT(flag == AbstractNonlinearSolver::ExitFlagType::converged);
        EXPECT(std::fabs(sol(0) - r0) < 1e-12);
        EXPECT(toy.evals == 1 + s.LastIterationCount());
        EXPECT(std::fabs(Jout(2, 2) - 1.0) < 1e-15 && Jout(0, 0) > 3.0);
        pars.tolerance = 1e-5;
    }
    {   // not converged within maxIterations -> flag, iteration count == maxIterations
        Toy toy;
        pars.maxIterations = 2;
        pars.tolerance = 1e-14;
        NewtonSolver s(&toy, &guess, &pars);
        arma::vec sol(3), hist;
        AbstractNonlinearSolver::ExitFlagType flag;
        s.Solve(sol, hist, flag);
        EXPECT(flag == AbstractNonlinearSolver::ExitFlagType::notConverged && s.LastIterationCount() == 2 && hist.n_elem == 3);
        pars.maxIterations = 10;
        pars.tolerance = 1e-5;
    }
    {   // damping 0.5 needs more iterations than damping 1
        Toy a, b;
        NewtonSolver sa(&a, &guess, &pars);
        arma::vec sol(3), hist;
        AbstractNonlinearSolver::ExitFlagType flag;
        sa.Solve(sol, hist, flag);
        const int full = sa.LastIterationCount();
        pars.damping = 0.5;
        pars.maxIterations = 60;
        NewtonSolver sb(&b, &guess, &pars);
        sb.Solve(sol, hist, flag);
        EXPECT(flag == AbstractNonlinearSolver::ExitFlagType::converged && sb.LastIterationCount() > full);
    }
    {   // eigenvalues: known spectra, complex pairs, defective and larger matrices
        auto sorted = [](std::vector<std::complex<double>> v) {
            std::sort(v.begin(), v.end(), [](const std::complex<double>& a, const std::complex<double>& b) {
                return a.real() != b.real() ? a.real() < b.real() : a.imag() < b.imag(); });
            return v;
        };
        const double rot[4] = {0, 1, -1, 0};                                // [[0,-1],[1,0]] column-major: +-i
        auto e = sorted(mi355::eig_general(rot, 2));
        EXPECT(std::abs(e[0] - std::complex<double>(0, -1)) < 1e-14 && std::abs(e[1] - std::complex<double>(0, 1)) < 1e-14);
        // companion matrix of (x-2)(x^2 - x + 0.5): eigenvalues 2, 0.5 +- 0.5i
        const double comp[9] = {0, 1, 0, 0, 0, 1, 1.0, -2.5, 3.0};           // columns; last column = -coefficients
        e = sorted(mi355::eig_general(comp, 3));
        EXPECT(std::abs(e[0] - std::complex<double>(0.5, -0.5)) < 1e-12 && std::abs(e[1] - std::complex<double>(0.5, 0.5)) < 1e-12 &&
               std::abs(e[2] - std::complex<double>(2.0, 0.0)) < 1e-12);
        const double jord[4] = {3, 0, 1, 3};                                   // Jordan block: double eigenvalue 3
        e = mi355::eig_general(jord, 2);
        EXPECT(std::abs(e[0] - 3.0) < 1e-7 && std::abs(e[1] - 3.0) < 1e-7);
        const int n = 6;                                                       // upper-triangular + similarity: spectrum 1..6
        std::vector<double> A(n * n, 0.0), T(n * n, 0.0), B(n * n, 0.0);
        for (int j = 0; j < n; ++j) for (int i = 0; i <= j; ++i) A[i + j * n] = (i == j) ? j + 1.0 : 0.3 * (i + 1) - 0.2 * j;
        for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) T[i + j * n] = (i == j) ? 1.0 : (i > j ? 0.5 / (1 + i - j) : 0.0);   // unit lower
        // B = T * A * T^-1 ; T^-1 by forward substitution on each unit vector
        std::vector<double> Ti(n * n, 0.0);
        for (int c2 = 0; c2 < n; ++c2) for (int i = 0; i < n; ++i) { double s2 = (i == c2); for (int k = 0; k < i; ++k) s2 -= T[i + k * n] * Ti[k + c2 * n]; Ti[i + c2 * n] = s2; }
        std::vector<double> TA(n * n, 0.0);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k) TA[i + j * n] += T[i + k * n] * A[k + j * n];
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k) B[i + j * n] += TA[i + k * n] * Ti[k + j * n];
        e = sorted(mi355::eig_general(B.data(), n));
        for (int i = 0; i < n; ++i) EXPECT(std::abs(e[i] - (double)(i + 1)) < 1e-9);
        // Stability counts: analytic Jacobian of the toy problem at its root; equationFree adds the identity
        Toy toy; ToyJac jac;
        arma::vec root(3);
        root(0) = r0; root(1) = r1; root(2) = std::exp(-r0);
        Stability smap(Stability::ProblemType::map, &toy, &jac), sflow(Stability::ProblemType::flow, &toy, &jac),
                  sef(Stability::ProblemType::equationFree, &toy, &jac), sfd(Stability::ProblemType::map, &toy);
        // J = [[2r0, 2r1, 0],[r1, r0, 0],[e^-r0, 0, 1]]: eigenvalues 1 and (3r0 +- sqrt(r0^2 + 8 r1^2))/2
        const double d = std::sqrt(r0 * r0 + 8 * r1 * r1), l1 = 0.5 * (3 * r0 + d), l2 = 0.5 * (3 * r0 - d);
        EXPECT(smap.ComputeNumUnstableEigenvalues(root) == (l1 > 1) + (l2 > 1));          // |1| is not > 1
        EXPECT(sflow.ComputeNumUnstableEigenvalues(root) == 3);
        EXPECT(sef.ComputeNumUnstableEigenvalues(root) == 3);                               // spectrum shifted by +1
        sfd.SetFiniteDifferenceEpsilon(1e-7);
        auto efd = sorted(sfd.ComputeEigenvalues(root));
        EXPECT(std::abs(efd[2].real() - l1) < 1e-5 && std::abs(efd[0].real() - std::min(1.0, l2)) < 1e-5);
    }
    ConvergenceCriterion c(1e-3);
    EXPECT(c.TestConvergence(1e-3) && !c.TestConvergence(1.0000001e-3));     // <=, ConvergenceCriterion.cpp:14
    std::printf(failures ? "host_selftest: %d FAILURES\n" : "host_selftest: all passed\n", failures);
    return failures ? 1 : 0;
}
