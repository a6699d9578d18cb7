"""Quick timing of the interp1 entry point for the bench's query sets (used while tuning kernels on the GPU box).
Optionally rebuilds the library first with MI_EXTRA_HIPCC_FLAGS=... (hipcc is present on the box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "default"
    if os.environ.get("MI_EXTRA_HIPCC_FLAGS") is not None:
        from armadillocudalinearinterpolation_amd import _build
        _build.build_lib(force=True)
    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import synth
    ctx = mi.Context(0)
    ng, nq = 1_000_000, 100_000_000
    X = np.arange(ng) / (ng - 1)
    Y = np.sin(2 * np.pi * X) + 0.5 * X
    u = synth.splitmix_uniform(0x5EED0003, nq, "cuda:0")
    su = torch.sort(u).values.contiguous()
    out = torch.empty_like(u)
    grids = {"general": mi.Grid1.from_nodes(ctx, X, Y), "uniform": mi.Grid1.uniform(ctx, 0.0, 1.0 / (ng - 1), Y)}
    Xj = (np.arange(ng) + 0.5 * np.random.default_rng(1).random(ng)) / ng
    grids["nonuniform"] = mi.Grid1.from_nodes(ctx, Xj, np.sin(Xj), sanitise=False)
    res = {}
    for gname, g in grids.items():
        for qname, q in (("random", u), ("sorted", su)):
            for _ in range(3):
                g.interp(q, out=out)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t = mi.Timer(ctx)
                t.start()
                for _ in range(5):
                    g.interp(q, out=out)
                t.stop()
                ts.append(t.elapsed_ms() / 5)
            res[gname + "_" + qname] = sorted(ts)[len(ts) // 2]
    print(tag, " ".join("%s %.4f" % kv for kv in res.items()), flush=True)


if __name__ == "__main__":
    main()
