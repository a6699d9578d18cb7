"""Region sweep vs streaming kernel across table sizes (random queries, closed-form table): validates the
table-size window in which launch_mode picks the sweep.  Run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import synth
    ctx = mi.Context(0)
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    u = synth.splitmix_uniform(0x5EED0003, nq, "cuda:0")
    out = torch.empty_like(u)
    for ng in (250_000, 400_000, 500_000, 700_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000, 16_000_000, 64_000_000):
        X = np.arange(ng) / (ng - 1)
        Y = np.sin(2 * np.pi * X) + 0.5 * X
        for kind in ("closed_form", "explicit"):
            if kind == "explicit":
                if ng > 16_000_000:
                    continue
                Xg = (np.arange(ng) + 0.5 * np.random.default_rng(1).random(ng)) / ng
                g = mi.Grid1.from_nodes(ctx, Xg, np.sin(Xg), sanitise=False)
            else:
                g = mi.Grid1.from_nodes(ctx, X, Y, sanitise=False)
            res = {}
            for name, hint in (("sweep", 1), ("stream", 2)):
                ctx.set_query_order(hint)
                for _ in range(2):
                    g.interp(u, out=out)
                t = mi.Timer(ctx)
                t.start()
                for _ in range(5):
                    g.interp(u, out=out)
                t.stop()
                ctx.synchronize()
                res[name] = t.elapsed_ms() / 5
            ctx.set_query_order(0)
            print("ng %9d %-11s table %7.1f MB mode %d : sweep %.4f ms  stream %.4f ms  ratio %.2f" % (
                ng, kind, g.info()["table_bytes"] / 1e6, g.info()["mode"], res["sweep"], res["stream"],
                res["stream"] / res["sweep"]), flush=True)
            del g


if __name__ == "__main__":
    main()
