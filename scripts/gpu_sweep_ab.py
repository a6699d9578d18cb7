#!/usr/bin/env python3
"""A-B of the pipelined region sweep's schedules (MI_SWEEP_SCHED values given on the command line, default "0 4"): kernel
time per launch on BASELINE configs[1], alternating, and a checksum of the outputs."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch
    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import synth
    ng, nq = 1_000_000, 100_000_000
    ctx = mi.Context(0)
    ctx.set_query_order(1)
    X, Y = synth.config_grid(ng)
    grid = mi.Grid1.from_nodes(ctx, X, Y, sanitise=False)
    dev = torch.device("cuda", 0)
    xq = synth.splitmix_uniform(0x5EED0003, nq, dev)
    yq = torch.empty_like(xq)
    for _ in range(3):
        grid.interp(xq, out=yq)
    torch.cuda.synchronize()
    tm = ctx.timer()
    res = []
    for _ in range(3):
        tm.start()
        for _ in range(10):
            grid.interp(xq, out=yq)
        tm.stop()
        res.append(tm.elapsed_ms() / 10)
    v = yq.view(torch.int64)
    print("%.4f %.4f %.4f ms  sum %d" % (res[0], res[1], res[2], int(v.sum().item())), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        scheds = sys.argv[1:] or ["0", "4"]
        for rep in range(2):
            for sc in scheds:
                env = dict(os.environ, MI_SWEEP_SCHED=sc)
                r = subprocess.run([sys.executable, "-u", os.path.abspath(__file__), "worker"], env=env, capture_output=True, text=True, timeout=300)
                print("SCHED %s : %s" % (sc, r.stdout.strip() or "FAILED " + r.stderr[-400:]), flush=True)
