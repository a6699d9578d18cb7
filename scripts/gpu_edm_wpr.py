"""Evolve kernel forms (1 / 4 waves per realisation; 16 was measured too and dropped) across realisation counts: timing + bit-equality of every
stage tap.  Run on the GPU box; used to place the thresholds in launch_evolve (mi_edm.hip)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import armadillocudalinearinterpolation_amd as mi
    ctx = mi.Context(0)
    Z = [0.3310, 0.6914, 1.3557]
    for n_grid in (1024, 512):
        for sigma in (0.0, 0.3):
            for R in (1, 8, 16, 32, 64, 128, 256, 300, 400, 512, 600, 1000, 2000, 4000, 16384):
                if sigma > 0 and R not in (64, 1000, 4000):
                    continue
                row, sig = [], None
                for wpr in (1, 4):
                    os.environ["MI_EDM_WAVES_PER_REALISATION"] = str(wpr)
                    edm = mi.EventDrivenMap(ctx, [13.0589], R, n_grid=n_grid, beta_stddev=sigma)
                    edm.ComputeF(Z)
                    f = edm.ComputeF(Z)
                    t = edm.last_timings()["evolve_ms"]
                    d = edm.debug_read()
                    h = hashlib.sha256(b"".join(np.ascontiguousarray(d[k]).tobytes() for k in ("t0", "i0", "t1", "i1", "accept")) + f.tobytes()).hexdigest()
                    if sig is None:
                        sig = h
                    row.append("%7.3f%s" % (t, "" if h == sig else " MISMATCH"))
                    edm.close()
                print("N %4d sigma %.1f R %6d : evolve ms  w1 %s  w4 %s" % (n_grid, sigma, R, row[0], row[1]), flush=True)
    os.environ.pop("MI_EDM_WAVES_PER_REALISATION", None)


if __name__ == "__main__":
    main()
