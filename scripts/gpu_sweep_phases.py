#!/usr/bin/env python3
"""Phase timing of the pipelined region-sweep kernel (mi_debug_sweep_timing hook): us per 16K-query tile between the
three workgroup barriers of a step (gather rounds | read-back | scatter + stores), for the gathering and the preparing
role, plus the kernel time.  BASELINE configs[1] inputs."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import armadillocudalinearinterpolation_amd as mi  # noqa: E402
from armadillocudalinearinterpolation_amd import _lib, synth  # noqa: E402

ng = int(os.environ.get("NG", 1_000_000))
nq = int(os.environ.get("NQ", 100_000_000))
ctx = mi.Context(0)
ctx.set_query_order(1)            # queries are unordered: no probe, always the region sweep
X, Y = synth.config_grid(ng)
grid = mi.Grid1.from_nodes(ctx, X, Y, sanitise=False)
dev = torch.device("cuda", 0)
xq = synth.splitmix_uniform(0x5EED0003, nq, dev)
yq = torch.empty_like(xq)
L = _lib.load()
cus = ctx.device_info()["compute_units"]
for _ in range(3):
    grid.interp(xq, out=yq)
torch.cuda.synchronize()
tm = ctx.timer()
tm.start()
for _ in range(10):
    grid.interp(xq, out=yq)
tm.stop()
print("kernel %.4f ms per launch (no stamps)" % (tm.elapsed_ms() / 10), flush=True)
ticks = torch.zeros(cus * 2 * 12, dtype=torch.int64, device=dev)
_lib.check(L.mi_debug_sweep_timing(ctx._h, C.c_void_p(ticks.data_ptr())), ctx._h)
grid.interp(xq, out=yq)
torch.cuda.synchronize()
tm.start()
grid.interp(xq, out=yq)
tm.stop()
print("kernel %.4f ms with stamps" % tm.elapsed_ms())
_lib.check(L.mi_debug_sweep_timing(ctx._h, None), ctx._h)
t = ticks.cpu().numpy().reshape(cus, 2, 2, 6).astype(np.float64) * 0.01          # us
ntiles = nq // 16384
per_group_tiles = ntiles / cus / 2.0                                               # tiles each group gathers (and prepares)
g = t[:, :, 0, :].mean(axis=(0, 1)) / per_group_tiles
pr = t[:, :, 1, :].mean(axis=(0, 1)) / per_group_tiles
print("gatherer : rounds %.2f | wait for preparer %.2f | read-back %.2f | stores (+loads) %.2f   = %.2f us per tile" %
      (g[3], g[0], g[1], g[2], g[[0, 1, 2, 3]].sum()))
print("preparer : loads issued %.2f | landed +%.2f | sorted +%.2f | wait for gatherer %.2f | clear %.2f | scatter %.2f   = %.2f us per tile" %
      (pr[3], pr[4], pr[5], pr[0], pr[1], pr[2], pr.sum()))
print("step = %.2f us -> %.3f ms per 1e8 queries" % (g[[0, 1, 2, 3]].sum(), g[[0, 1, 2, 3]].sum() * ntiles / cus * 1e-3))
