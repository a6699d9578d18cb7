"""Exploration microbench (not a test): how does interp1 time depend on the table footprint the
random queries touch?  Queries are uniform in [0, frac) of the table."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from armadillocudalinearinterpolation_amd import _lib  # noqa: E402

_lib.load(strict=False)
import armadillocudalinearinterpolation_amd as mi  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = mi.Context(0)
dev = torch.device("cuda:0")
NG, NQ = 10**6, 10**8
X6 = np.arange(NG) / (NG - 1)
Y6 = np.sin(2 * np.pi * X6) + 0.5 * X6
base = torch.rand(NQ, dtype=torch.float64, device=dev)
out = torch.empty_like(base)
tm = ctx.timer()
grids = {"mode1": mi.Grid1.from_nodes(ctx, X6, Y6, sanitise=False),
         "mode0": mi.Grid1.uniform(ctx, 0.0, 1.0 / (NG - 1), Y6)}
res = {}
for gname, grid in grids.items():
    if mode not in ("all", gname):
        continue
    for frac in (1.0, 0.5, 0.25, 0.125, 1.0 / 64):
        q = base * frac
        grid.interp(q, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            tm.start(); grid.interp(q, out=out); tm.stop()
            ts.append(tm.elapsed_ms())
        ms = float(np.median(ts))
        res["%s_frac%.4f" % (gname, frac)] = ms
        print(gname, "table fraction %.4f -> %.4f ms  (%.1f%% of 8 TB/s)" % (frac, ms, 16 * NQ / ms / 1e6 / 80), flush=True)
# plain copy for reference (same bytes)
ts = []
for _ in range(reps):
    tm.start(); out.copy_(base); tm.stop(); ts.append(tm.elapsed_ms())
res["torch_copy"] = float(np.median(ts))
print("torch copy", res["torch_copy"])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/explore_%s.json" % mode, "w"), indent=1)
