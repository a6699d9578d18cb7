"""Random / sorted queries over small closed-form tables: LDS-table kernel (default) vs streaming kernel (ORDERED hint)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import synth
    ctx = mi.Context(0)
    nq = 100_000_000
    u = synth.splitmix_uniform(0x5EED0003, nq, "cuda:0")
    su = torch.sort(u).values.contiguous()
    out = torch.empty_like(u)
    cases = [(ng, "closed") for ng in (100, 1000, 10_000, 16_000, 20_000, 100_000)] + [(ng, "jitter") for ng in (1000, 4000, 8000, 10_000, 100_000)]
    for ng, kind in cases:
        X = np.arange(ng) / (ng - 1) if kind == "closed" else (np.arange(ng) + 0.5 * np.random.default_rng(ng).random(ng)) / ng
        g = mi.Grid1.from_nodes(ctx, X, np.sin(2 * np.pi * X) + 0.5 * X, sanitise=False)
        row = []
        for qname, q in (("random", u), ("sorted", su)):
            for hint in (0, 2):
                ctx.set_query_order(hint)
                for _ in range(2):
                    g.interp(q, out=out)
                t = mi.Timer(ctx)
                t.start()
                for _ in range(5):
                    g.interp(q, out=out)
                t.stop()
                ctx.synchronize()
                row.append("%s/%s %.4f" % (qname, "auto" if hint == 0 else "stream", t.elapsed_ms() / 5))
        ctx.set_query_order(0)
        print("ng %7d %-6s mode %d : %s" % (ng, kind, g.info()["mode"], "  ".join(row)), flush=True)


if __name__ == "__main__":
    main()
