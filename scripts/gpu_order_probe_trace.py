"""Per-call times of mi_interp1_f64_dev with the query order left to the device-side probe (AUTO), across changes of the
query set: how many calls after a change still run the kernel predicted for the previous set?  (not a test)"""
import os
import sys

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
for name, q in (("random", xr), ("sorted", xs), ("uniform", xu), ("random", xr), ("uniform", xu), ("sorted", xs)):
    ts = []
    for _ in range(8):
        tm.start()
        grid.interp(q, out=out)
        tm.stop()
        torch.cuda.synchronize()
        ts.append(tm.elapsed_ms())
    print("%-8s" % name, " ".join("%.3f" % t for t in ts), flush=True)
