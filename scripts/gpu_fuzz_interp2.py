"""Differential fuzz of the scattered bilinear path (mi_interp2_f64_dev) against the CPU oracle (GPU box only)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle


def axis(rng, n):
    kind = rng.integers(4)
    i = np.arange(n, dtype=np.float64)
    if kind == 0:
        return np.linspace(rng.uniform(-5, 0), rng.uniform(0.5, 5), n)
    if kind == 1:
        return rng.uniform(-2, 2) + rng.uniform(1e-3, 1.0) * (i + rng.uniform(0.1, 0.9) * rng.random(n))
    if kind == 2:
        return np.cumsum(rng.random(n) ** 3 + 1e-6)
    return np.unique((i / max(n - 1, 1)) ** rng.uniform(1.5, 3.0) * rng.uniform(1, 9))


def run(budget, seed, ctx=None):
    import armadillocudalinearinterpolation_amd as mi
    rng = np.random.default_rng(seed)
    ctx = ctx or mi.Context(0)
    t0, cases, scalar = time.time(), 0, 0
    beat = t0
    while time.time() - t0 < budget:
        if time.time() - beat > 60.0:          # a sign of life every minute (a silent GPU command is taken to be hung)
            beat = time.time()
            print("  ... %d cases after %.0f s" % (cases, beat - t0), flush=True)
        nx, ny = (int(rng.choice([2, 3, 17, 64, 255, 700])) for _ in range(2))
        uniform = rng.random() < 0.35
        if uniform:
            x0, dx, y0, dy = rng.uniform(-3, 3), rng.uniform(1e-3, 2), rng.uniform(-3, 3), rng.uniform(1e-3, 2)
            xg, yg = x0 + dx * np.arange(nx), y0 + dy * np.arange(ny)
        else:
            xg, yg = axis(rng, nx), axis(rng, ny)
            nx, ny = xg.size, yg.size
            if nx < 2 or ny < 2:
                continue
        Z = rng.standard_normal((ny, nx)) * rng.uniform(0.1, 100)        # Z[row = y index, col = x index]
        nq = int(rng.choice([1, 3, 1000, 65537, 400001]))
        xq = rng.random(nq) * (xg[-1] - xg[0]) * 1.04 + xg[0] - 0.02 * (xg[-1] - xg[0])
        yq = rng.random(nq) * (yg[-1] - yg[0]) * 1.04 + yg[0] - 0.02 * (yg[-1] - yg[0])
        if nq > 8:
            pick = rng.integers(0, nq, nq // 4)
            xq[pick] = xg[rng.integers(0, nx, pick.size)]                   # exactly on grid lines
            pick = rng.integers(0, nq, nq // 4)
            yq[pick] = yg[rng.integers(0, ny, pick.size)]
            xq[:4] = [np.nan, xg[0], xg[-1], np.inf]
            yq[:4] = [yg[0], np.nan, yg[-1], yg[0]]
        compact = bool(rng.integers(2))
        if uniform:
            g = mi.Grid2.uniform(ctx, x0, dx, nx, y0, dy, ny, Z, compact=compact)
            ref = oracle.interp2_bilinear_uniform(x0, dx, nx, y0, dy, ny, Z, xq, yq, extrap=-9.5, nthreads=8)
        else:
            g = mi.Grid2.from_axes(ctx, xg, yg, Z, compact=compact)
            ref = oracle.interp2_bilinear(xg, yg, Z, xq, yq, extrap=-9.5, nthreads=8)
        txq, tyq = torch.from_numpy(xq).cuda(), torch.from_numpy(yq).cuda()
        out = g.interp(txq, tyq, extrap=-9.5).cpu().numpy()
        # a second pass through the 8-byte-aligned (scalar) kernel: same bits
        if nq > 8 and rng.integers(4) == 0:
            out2 = g.interp(txq[1:], tyq[1:], extrap=-9.5).cpu().numpy()
            scalar += 1
            if not np.array_equal(out2, out[1:], equal_nan=True):
                print("SCALAR KERNEL MISMATCH", uniform, compact, nx, ny, nq, flush=True)
                raise AssertionError("scalar kernel differs from the vector kernel")
        if not np.array_equal(out, ref, equal_nan=True):
            bad = np.flatnonzero(~((out == ref) | (np.isnan(out) & np.isnan(ref))))
            print("MISMATCH", uniform, compact, nx, ny, nq, bad[:5], out[bad[:5]], ref[bad[:5]], flush=True)
            raise AssertionError("differential fuzz mismatch (details printed above)")
        cases += 1
        del g
    print("interp2 fuzz ok: %d cases (%d of them also through the scalar kernel) in %.0f s" % (cases, scalar, time.time() - t0), flush=True)
    return {"cases": cases, "scalar": scalar}


def main():
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)


if __name__ == "__main__":
    main()
