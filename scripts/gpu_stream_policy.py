#!/usr/bin/env python3
"""Cache-policy bits on the query/result streams of the region-sweep kernels (build hook MI_STREAM_LD / MI_STREAM_ST in
csrc/mi_interp1.hip): kernel time per launch for the pipelined and the simple form (closed-form grid, 8 MB table) and
for the explicit {x,y} grid (16 MB table), with a checksum of the outputs against the shipped build.

Driver (no argument): for every scripts/_variants/lib_*.so (built beforehand, see profiles/r02_exp_stream_cache_policy.log)
copy it over the package's library ON THE GPU BOX'S SCRATCH COPY and run this file as a worker in a child process.
"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import numpy as np
    import torch
    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import synth
    ng, nq = 1_000_000, 100_000_000
    ctx = mi.Context(0)
    ctx.set_query_order(1)
    X, Y = synth.config_grid(ng)
    dev = torch.device("cuda", 0)
    xq = synth.splitmix_uniform(0x5EED0003, nq, dev)
    yq = torch.empty_like(xq)
    un = synth.splitmix_uniform(0x5EED0002, ng, torch.device("cpu")).numpy()
    Xn = (np.arange(ng) + 0.5 * un) / ng
    out = []
    for name, xs in (("closed-form", X), ("explicit", Xn)):
        grid = mi.Grid1.from_nodes(ctx, xs, Y, sanitise=False)
        for _ in range(3):
            grid.interp(xq, out=yq)
        torch.cuda.synchronize()
        tm = ctx.timer()
        tm.start()
        for _ in range(10):
            grid.interp(xq, out=yq)
        tm.stop()
        v = yq.view(torch.int64)
        out.append("%s %.4f ms sum %d xor-fold %d" % (name, tm.elapsed_ms() / 10, int(v.sum().item()),
                                                       int((v ^ (v >> 17)).sum().item())))
        del grid
    print(" | ".join(out), flush=True)


def main():
    lib = os.path.join(ROOT, "armadillocudalinearinterpolation_amd", "libmi355interp.so")
    keep = lib + ".shipped"
    shutil.copy(lib, keep)
    variants = sorted(glob.glob(os.path.join(ROOT, "scripts", "_variants", "lib_*.so")))
    try:
        for rep in range(2):
            for v in variants:
                shutil.copy(v, lib)
                for form in ("2", "1"):
                    env = dict(os.environ, MI_SWEEP_VARIANT=form)
                    r = subprocess.run([sys.executable, "-u", os.path.abspath(__file__), "worker"], env=env,
                                       capture_output=True, text=True, timeout=300)
                    tag = os.path.basename(v)[4:-3]
                    print("%-12s form %s : %s" % (tag, "pipelined" if form == "2" else "simple   ",
                                                  r.stdout.strip() or ("FAILED " + r.stderr.strip()[-300:])), flush=True)
    finally:
        shutil.copy(keep, lib)
        os.remove(keep)


if __name__ == "__main__":
    worker() if len(sys.argv) > 1 else main()
