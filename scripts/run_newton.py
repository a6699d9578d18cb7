#!/usr/bin/env python3
"""BASELINE configs 4-5: EventDrivenMap residual / full Newton solve with realisations sharded over the GPUs of
one node (one process per GPU; all-reduce of the 2S+1 fp64 scalars of the partial block per residual evaluation).

  python scripts/run_newton.py --real 125000                       # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/run_newton.py --real 1000000
Options: --threads N (grid points, default 1024) --fast --residual-only --backend nccl|gloo --one-device
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--real", type=int, default=1_000_000, help="total realisations over all ranks")
    ap.add_argument("--threads", type=int, default=1024)
    ap.add_argument("--fast", action="store_true")
    ap.add_argument("--sigma", type=float, default=0.0)
    ap.add_argument("--residual-only", action="store_true", help="config 4: time ComputeF(Z0) only")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank on cuda:0 (gloo only)")
    ap.add_argument("--true-mean", action="store_true", help="mean_quirk = 0 (default: the reference's averaging, "
                    "EventDrivenMap.cu:800-824)")
    ap.add_argument("--max-iterations", type=int, default=10)
    a = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import newton, sharding

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if a.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    ctx = mi.Context(local)
    lo, hi = sharding.shard_bounds(a.real, rank, world)
    edm = mi.EventDrivenMap(ctx, [13.0589], hi - lo, n_grid=a.threads, real_offset=lo, beta_stddev=a.sigma,
                            math_mode=mi.MATH_FAST if a.fast else mi.MATH_EXACT, seed=0x5EED0005,
                            mean_quirk=0 if a.true_mean else 1)
    ms = []

    class Sharded:
        """ComputeF = local partial sums on this GPU -> all-reduce(sum) of the 2S+1-double partial block -> host epilogue."""

        def ComputeF(self, Z):
            t = time.perf_counter()
            _, part = edm.ComputeF(Z, want_partial=True)
            tt = torch.from_numpy(part.copy())
            if world > 1:
                tt = tt.to(dev) if a.backend == "nccl" else tt
                dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            f = edm.residual_from_sums(Z, tt.cpu().numpy())
            ms.append((time.perf_counter() - t) * 1e3)
            return f

    Z0 = [float(np.float32(0.3310)), float(np.float32(0.6914)), float(np.float32(1.3557))]      # Driver.cu:24
    prob = Sharded()
    prob.ComputeF(Z0)                                                                                # warm-up
    ms.clear()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if a.residual_only:
        f = prob.ComputeF(Z0)
        out = {"config": "EventDrivenMap residual eval (BASELINE configs[3])", "f": f.tolist()}
    else:
        pars = newton.ParameterList(tolerance=1e-4, maxIterations=a.max_iterations, printOutput=(rank == 0), finiteDifferenceEpsilon=1e-2)
        u, hist, conv, it = newton.NewtonSolver(prob, Z0, pars, printer=lambda s: print(s, file=sys.stderr)).Solve()
        out = {"config": "Full NewtonSolver loop, Driver.cu problem (BASELINE configs[4])", "converged": bool(conv),
               "iterations": it, "solution": u.tolist(), "history": hist, "final_norm": hist[-1]}
    wall = time.perf_counter() - t0
    if rank == 0:
        out.update({"n_gpus": world, "realisations_total": a.real, "realisations_per_gpu": hi - lo, "grid_points": a.threads,
                    "math": "fast" if a.fast else "exact", "sigma": a.sigma, "wall_s": wall, "compute_f_calls": len(ms),
                    "compute_f_ms_mean": float(np.mean(ms)), "evolve_ms_last": edm.last_timings()["evolve_ms"],
                    "mean": "true" if a.true_mean else "reference",
                    "exchange": "all-reduce(sum) of %d fp64 scalars per ComputeF" % (2 * int(edm.params.n_spikes) + 1)})
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
