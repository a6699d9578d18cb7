import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import armadillocudalinearinterpolation_amd as mi
from armadillocudalinearinterpolation_amd import newton
ctx = mi.Context(0)
Z = np.array([0.3310, 0.6914, 1.3557])
for n_grid, R in ((512, 1000), (1024, 1000), (1024, 300), (1024, 5000), (1024, 20000)):
    for concurrent in (False, True):
        prob = mi.EventDrivenMap(ctx, [13.0589], R, n_grid=n_grid)
        pars = newton.ParameterList(tolerance=1e-4, maxIterations=10, printOutput=False, finiteDifferenceEpsilon=1e-2)
        solver = newton.NewtonSolver(prob, Z.copy(), pars)
        solver.concurrent_columns = concurrent
        prob.ComputeF(Z); prob.ComputeFBatch([Z, Z, Z]) if concurrent else None
        t = time.perf_counter(); sol, hist, conv, its = solver.Solve(); dt = time.perf_counter() - t
        print("N %4d R %6d concurrent columns %d : Newton solve %.1f ms, %d iterations, %d evaluations, converged %s" % (n_grid, R, concurrent, dt * 1e3, its, solver.evaluations, conv), flush=True)
        prob.close()
