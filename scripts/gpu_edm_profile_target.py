"""Single-configuration ComputeF run for rocprofv3 (EXACT math, N = 1024, R = 32768)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armadillocudalinearinterpolation_amd as mi
ctx = mi.Context(0)
mode = mi.MATH_FAST if len(sys.argv) > 1 and sys.argv[1] == "fast" else mi.MATH_EXACT
edm = mi.EventDrivenMap(ctx, [13.0589], 32768, n_grid=1024, math_mode=mode)
for _ in range(3):
    edm.ComputeF([0.3310, 0.6914, 1.3557])
print(edm.last_timings())
