"""Python mirror of the host-side Newton loop (NewtonSolver.cpp:40-197 of the reference; the C++ form of the
same logic is host/newton_solver.cpp).  It exists for multi-GPU runs: every rank (one process per GPU) runs
this tiny replicated loop and only the residual evaluation is sharded (sharding.ShardedResidual).

problem: any object with ComputeF(u) -> f (numpy float64) and optionally PostProcess().
"""
import numpy as np


class ParameterList:
    """NewtonSolver::ParameterList defaults (NewtonSolver.hpp:19-26)."""

    def __init__(self, tolerance=1e-5, maxIterations=10, printOutput=True, finiteDifferenceEpsilon=1e-8, damping=1.0):
        self.tolerance = tolerance
        self.maxIterations = maxIterations
        self.printOutput = printOutput
        self.finiteDifferenceEpsilon = finiteDifferenceEpsilon
        self.damping = damping


class NewtonSolver:
    def __init__(self, problem, initial_guess, pars, jacobian=None, printer=print):
        self.problem, self.guess, self.pars, self.jacobian, self._print = problem, initial_guess, pars, jacobian, printer
        self.evaluations = 0
        self.concurrent_columns = True   # use problem.ComputeFBatch for the finite-difference columns when it exists

    def _F(self, u):
        self.evaluations += 1
        return np.asarray(self.problem.ComputeF(u), dtype=np.float64)

    def _fd_jacobian(self, u, f):
        n, eps = u.size, self.pars.finiteDifferenceEpsilon
        inv_eps = eps ** -1
        J = np.empty((n, n))
        batch = getattr(self.problem, "ComputeFBatch", None)
        if batch is not None and self.concurrent_columns:
            # the n perturbed residuals are independent: enqueue them all, then collect (same values, same order of
            # floating-point operations per column as the loop below)
            pert = []
            for i in range(n):
                du = u.copy()
                du[i] += eps
                pert.append(du)
            F = np.asarray(batch(pert), dtype=np.float64)
            self.evaluations += n
            for i in range(n):
                J[:, i] = (F[i] - f) * inv_eps
            return J
        du = u.copy()
        for i in range(n):
            if i > 0:
                du[i - 1] = u[i - 1]
            du[i] += eps
            J[:, i] = (self._F(du) - f) * inv_eps
        return J

    def Solve(self):
        """Returns (solution, residual_history, converged, iterations)."""
        p = self.pars
        if p.printOutput:
            self._print("-" * 48 + "\n Attempt to solve nonlinear problem with Newton Method\n max number of iterations = %d\n"
                        " tolerance = %g\n" % (p.maxIterations, p.tolerance) + "-" * 48)
        u = np.array(self.guess, dtype=np.float64)
        self.evaluations = 0
        f = self._F(u)
        hist = [float(np.linalg.norm(f))]
        if p.printOutput:
            self._print("%10s%25s\n%10d%25.6e" % ("Iteration", "error estimate", 0, hist[0]))
        it, converged = 0, hist[0] <= p.tolerance
        while it < p.maxIterations and not converged:
            J = self.jacobian(u) if self.jacobian else self._fd_jacobian(u, f)
            u = u + p.damping * np.linalg.solve(J, -f)
            it += 1
            f = self._F(u)
            hist.append(float(np.linalg.norm(f)))
            converged = hist[-1] <= p.tolerance
            if p.printOutput:
                self._print("%10d%25.6e" % (it, hist[-1]))
        if hasattr(self.problem, "PostProcess"):
            self.problem.PostProcess()
        if p.printOutput:
            self._print("-" * 48 + "\nThe method %s after %d iterations" % ("converged" if converged else "failed to converge", it))
        return u, hist, converged, it
