"""Per-call ms of ordered query sets after a run of random ones, with different idle times in between (not a test)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armadillocudalinearinterpolation_amd as mi  # noqa: E402
from armadillocudalinearinterpolation_amd import synth  # noqa: E402

ctx = mi.Context(0)
dev = torch.device("cuda", 0)
NG, NQ = 1_000_000, 100_000_000
X, Y = synth.config_grid(NG)
grid = mi.Grid1.from_nodes(ctx, X, Y, sanitise=False)
xr = synth.splitmix_uniform(0x5EED0003, NQ, dev)
xs = torch.sort(xr).values
xu = torch.arange(NQ, dtype=torch.float64, device=dev) / (NQ - 1)
out = torch.empty_like(xr)
tm = ctx.timer()


def run(q, n):
    ts = []
    for _ in range(n):
        tm.start()
        grid.interp(q, out=out)
        tm.stop()
        torch.cuda.synchronize()
        ts.append(tm.elapsed_ms())
    return ts


for idle in (0.0, 0.05, 1.0, 3.0):
    run(xr, 25)
    for name, q in (("sorted", xs), ("uniform", xu), ("sorted", xs), ("uniform", xu)):
        torch.cuda.synchronize()
        time.sleep(idle)
        ts = run(q, 14)
        print("idle %.2f s %-8s" % (idle, name), " ".join("%.3f" % t for t in ts), flush=True)

# how long does the dip after an idle last?
torch.cuda.synchronize()
time.sleep(0.05)
ts = run(xs, 120)
print("after 50 ms idle, 120 sorted calls:", " ".join("%.3f" % t for t in ts), flush=True)
run(xr, 25)
ts = run(xs, 120)
print("right after 25 random calls, 120 sorted calls:", " ".join("%.3f" % t for t in ts), flush=True)
