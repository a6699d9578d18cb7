"""Differential fuzz of the EventDrivenMap pipeline (EXACT math) against oracle/edm_oracle.c (run on the GPU box;
not part of the test suite).  Random model parameters, grid sizes, spike counts, heterogeneity, both evolve kernel
forms and, in three cases of ten, a launch of more realisations than the device holds at once; every stage
tap must be bit-identical."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle


def run(budget, seed, ctx=None):
    import armadillocudalinearinterpolation_amd as mi
    rng = np.random.default_rng(seed)
    ctx = ctx or mi.Context(0)
    t0, cases, accepted, filled = time.time(), 0, 0, 0
    beat = t0
    while time.time() - t0 < budget:
        if time.time() - beat > 60.0:          # a sign of life every minute (a silent GPU command is taken to be hung)
            beat = time.time()
            print("  ... %d cases after %.0f s" % (cases, beat - t0), flush=True)
        S = int(rng.choice([1, 2, 3, 3, 3, 4, 5]))
        kw = dict(
            n_grid=int(rng.choice([64, 100, 256, 500, 512, 1000, 1024])), n_real=int(rng.choice([1, 2, 3, 5, 9])),
            n_spikes=S, beta_mean=float(np.float32(rng.uniform(6.0, 20.0))), beta_stddev=float(np.float32(rng.choice([0.0, 0.0, 0.2, 1.0]))),
            a1=float(np.float32(rng.uniform(8, 14))), a2=float(np.float32(rng.uniform(5, 9))), b1=float(np.float32(rng.uniform(4, 6))),
            b2=float(np.float32(rng.uniform(3, 4))), I=float(np.float32(rng.uniform(0.8, 0.97))), time_horizon=float(np.float32(rng.uniform(1.0, 6.0))),
            seed=int(rng.integers(1, 2**40)), max_events=3000, mean_quirk=int(rng.integers(0, 2)), real_offset=int(rng.integers(0, 1000)),
        )
        c = rng.uniform(0.2, 0.5)
        Z = np.concatenate([[c], np.sort(rng.uniform(0.3, 2.5, S - 1))]) if S > 1 else np.array([c])
        if rng.random() < 0.3:
            # A launch of more realisations than the device holds at once (workgroups in several rounds, the automatic kernel
            # choice lands on the throughput form with the exact uniform-divisor quotient): 3200 identical realisations against ONE
            # oracle realisation.  Some with a threshold gap outside (0, 1] and some with a far field that decays to tiny values.
            kw.update(n_real=3200, beta_stddev=0.0, n_grid=int(rng.choice([64, 256, 512, 992, 1000, 1024])), max_events=1500)
            if rng.random() < 0.25:
                kw["I"] = float(np.float32(rng.uniform(0.5, 1.3)))
            if rng.random() < 0.25:
                kw["b1"], kw["b2"] = float(np.float32(4.0 * kw["b1"])), float(np.float32(4.0 * kw["b2"]))
            os.environ["MI_EDM_WAVES_PER_REALISATION"] = "1"
            R = kw.pop("n_real")
            edm = mi.EventDrivenMap(ctx, [kw.pop("beta_mean")], R, **kw)
            edm.ComputeF(Z)
            dbg = edm.debug_read()
            _, d = oracle.edm_compute_f(oracle.edm_default_params(beta_mean=edm.params.beta_mean, n_real=1, **kw), Z)
            bad = [k for k in ("seed_ind", "w", "v", "s") if not np.array_equal(dbg[k], d[k], equal_nan=True)]
            bad += [k for k in ("t0", "i0", "t1", "i1")
                    if not np.array_equal(np.asarray(dbg[k]).reshape(S, R), np.repeat(np.asarray(d[k]).reshape(S, 1), R, axis=1), equal_nan=True)]
            if bad or not np.array_equal(np.asarray(dbg["accept"]), np.repeat(np.asarray(d["accept"]), R)):
                print("MISMATCH (filled device)", bad, kw, Z, flush=True)
                raise AssertionError("differential fuzz mismatch (details printed above)")
            accepted += int(d["accept"][0])
            cases += 1
            filled += 1
            edm.close()
            continue
        os.environ["MI_EDM_WAVES_PER_REALISATION"] = str(rng.choice([1, 4]))
        edm = mi.EventDrivenMap(ctx, [kw.pop("beta_mean")], kw.pop("n_real"), **kw)
        f, partial = edm.ComputeF(Z, want_partial=True)
        dbg = edm.debug_read()
        p = oracle.edm_default_params(beta_mean=edm.params.beta_mean, n_real=edm.params.n_real, **kw)
        fo, d = oracle.edm_compute_f(p, Z, nthreads=8)
        bad = [k for k in ("seed_ind", "w", "v", "s", "t0", "i0", "t1", "i1", "accept", "restricted") if not np.array_equal(dbg[k], d[k], equal_nan=True)]
        if bad or partial[S] != d["sums"][S] or not np.allclose(f, fo, rtol=0, atol=3e-7, equal_nan=True):
            print("MISMATCH", bad, kw, Z, f, fo, flush=True)
            raise AssertionError("differential fuzz mismatch (details printed above)")
        accepted += int(d["sums"][S] > 0)
        cases += 1
        edm.close()
    os.environ.pop("MI_EDM_WAVES_PER_REALISATION", None)
    print("edm fuzz ok: %d cases in %.0f s (%d with accepted realisations, %d of them device-filling launches)"
          % (cases, time.time() - t0, accepted, filled), flush=True)
    return {"cases": cases, "accepted": accepted, "filled": filled}


def main():
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)


if __name__ == "__main__":
    main()
