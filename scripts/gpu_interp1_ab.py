"""A/B timing of the headline call (1e8 random queries, 1e6-node table) with different builds of the library on ONE box (not a
test).  Usage: gpu_interp1_ab.py <lib.so> [<lib.so> ...]   -- each library in its own process (bench.py --no-extra), three passes"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = ("import sys; sys.path.insert(0, %r); import armadillocudalinearinterpolation_amd._build as b; b.LIB_PATH = sys.argv[1]; "
        "b.is_stale = lambda: False; sys.argv = ['bench.py', '--no-extra', '--no-cpu-baseline', '--steps', '40', '--warmup', '10'] + sys.argv[2:]; "
        "__file__ = %r; exec(open(__file__).read())" % (ROOT, os.path.join(ROOT, "bench.py")))
for rep in range(3):
    for lib in sys.argv[1:]:
        for extra in ([], ["--table", "nonuniform"]):
            out = subprocess.run([sys.executable, "-c", SHIM, os.path.abspath(lib)] + extra, capture_output=True, text=True, timeout=300)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(lib, "FAILED", out.stderr[-400:], flush=True)
                continue
            d = json.loads(line[-1])
            print("%-22s pass %d %-12s %.4f ms  kernel %s" % (os.path.basename(lib), rep, " ".join(extra) or "general", d["ms_per_step"],
                                                                d["roofline"].get("kernel_ms")), flush=True)
