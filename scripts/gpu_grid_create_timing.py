import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import armadillocudalinearinterpolation_amd as mi
ctx = mi.Context(0)
n = 10**6
cases = {"linspace": np.linspace(0, 1, n), "i/(n-1)": np.arange(n) / (n - 1), "jitter": (np.arange(n) + 0.5 * np.random.default_rng(1).random(n)) / n,
         "clustered": np.unique(np.sort(np.random.default_rng(2).random(n) ** 3)), "shuffled linspace": np.random.default_rng(3).permutation(np.linspace(0, 1, n))}
for name, X in cases.items():
    Y = np.sin(X)
    for sanitise in (True, False):
        if name.startswith("shuffled") and not sanitise:
            continue
        ts = []
        for _ in range(3):
            t = time.perf_counter(); g = mi.Grid1.from_nodes(ctx, X, Y, sanitise=sanitise); ts.append(time.perf_counter() - t); m = g.info()["mode"]; g.close()
        print("%-18s sanitise=%d mode %d : create %.1f ms" % (name, sanitise, m, min(ts) * 1e3), flush=True)
