"""Why is the uniform query set XI_j = j/(NQ-1) slower than the same number of sorted random queries in the streaming
kernel?  Times a few ordered query sets over the headline table (not a test)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armadillocudalinearinterpolation_amd as mi  # noqa: E402
from armadillocudalinearinterpolation_amd import synth  # noqa: E402

ctx = mi.Context(0)
dev = torch.device("cuda", 0)
NG, NQ = 1_000_000, 100_000_000
X, Y = synth.config_grid(NG)
grid = mi.Grid1.from_nodes(ctx, X, Y, sanitise=False)
gu = mi.Grid1.uniform(ctx, 0.0, 1.0 / (NG - 1), Y)
j = torch.arange(NQ, dtype=torch.float64, device=dev)
sets = {
    "sorted random": torch.sort(synth.splitmix_uniform(0x5EED0003, NQ, dev)).values,
    "j/(NQ-1)": j / (NQ - 1),
    "(j+0.5)/NQ": (j + 0.5) / NQ,
    "j/(NQ-1) * 0.999999": j / (NQ - 1) * 0.999999,
    "j/(NQ-1) + jitter 3e-9": (j / (NQ - 1) + (synth.splitmix_uniform(7, NQ, dev) - 0.5) * 3e-9).clamp(0, 1),
    "sorted random, every value repeated 4x": torch.sort(synth.splitmix_uniform(9, NQ // 4, dev)).values.repeat_interleave(4),
}
del j
out = torch.empty(NQ, dtype=torch.float64, device=dev)
ctx.set_query_order(2)          # declared ordered: the streaming kernel, no probe
for rep, (gname, g) in enumerate((("warm-up pass", grid), ("explicit X (mode 0 detected)", grid), ("implicit uniform", gu), ("explicit X again", grid))):
    for name, q in (list(sets.items()) if rep != 3 else list(sets.items())[::-1]):
        for _ in range(3):
            g.interp(q, out=out)
        torch.cuda.synchronize()
        tm = ctx.timer()
        tm.start()
        for _ in range(10):
            g.interp(q, out=out)
        tm.stop()
        torch.cuda.synchronize()
        print("%-30s %-40s %.4f ms" % (gname, name, tm.elapsed_ms() / 10), flush=True)
