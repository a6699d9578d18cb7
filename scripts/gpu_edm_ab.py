"""A/B timing of Evolve with different builds of the library on ONE box (not a test).
Usage: gpu_edm_ab.py <lib.so> [<lib.so> ...]   -- each library is loaded in its own process that runs scripts/gpu_edm_timing_short.py
(or the script named by MI_AB_SCRIPT, e.g. gpu_edm_timing_small.py)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = ("import sys; sys.path.insert(0, %r); import armadillocudalinearinterpolation_amd._build as b; b.LIB_PATH = sys.argv[1]; "
        "b.is_stale = lambda: False; sys.argv = ['x', '3']; __file__ = %r; exec(open(__file__).read())" % (ROOT, os.path.join(ROOT, "scripts", os.environ.get("MI_AB_SCRIPT", "gpu_edm_timing_short.py"))))
for rep in range(2):
    for lib in sys.argv[1:]:
        print("== %s (pass %d)" % (os.path.basename(lib), rep), flush=True)
        out = subprocess.run([sys.executable, "-c", SHIM, os.path.abspath(lib)], capture_output=True, text=True, timeout=300)
        print("".join(l + "\n" for l in out.stdout.splitlines() if "exact" in l or "Error" in l), end="", flush=True)
        if out.returncode:
            print(out.stderr[-800:], flush=True)
