#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X interpolation hot path.

Workload (BASELINE.json configs[1]): 1-D linear interpolation of 1e8 random fp64 queries (SplitMix64 seed
0x5EED0003, U[0,1)) over a 1e6-node table X_i = i/(NG-1), Y_i = sin(2 pi X_i) + 0.5 X_i, through the
arma::interp1-shaped entry point (explicit X, general table).  One "step" = one pass of
mi_interp1_f64_dev over the rank's 1e8 resident queries.  Inputs are generated in HBM before timing.

  python bench.py --gpus N --steps K --warmup W
  N > 1, either launched by the driver as
      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N
  or invoked plainly (no WORLD_SIZE in the environment): this process then touches neither torch nor the GPU, starts that
  same launcher as a CHILD process (one rank per GPU), relays its output and exits with its code.
  --backend group: ONE process drives the N GPUs through the C ABI's mi_group_* entry points (csrc/mi_group.hip: one
  context + stream per device, RCCL bound by dlopen); the JSON line gains "rccl_ranks", the size of the ncclCommInitAll
  communicator as RCCL reports it.

Multi-GPU: the query axis shards trivially: every rank owns its own 1e8-query shard and a
replica of the table; there is no data-path collective in the timed region ("scaling": "weak").  The optional
all-gather that reassembles the full result vector on every rank is timed separately (extra.allgather_ms).

What to expect from the headline, up front (profiles/r04_bench_rocprof_summary.json,
profiles/r03_exp_gather_rate_vs_active_cus.log): uniformly random queries over a table larger than an XCD's L2 are bound by
ONE L2 REQUEST PER QUERY -- an XCD's vector request path takes about 33 per ns, so the launch's 115.0 M requests need at
least 0.44 ms: 46 % of 8 TB/s is the ceiling of any single-pass design on this metric, and the shipped kernel measures
0.68 ms = 29.4 % (roofline.request_floor_ms_at_measured_cap carries the floor in every line).  north_star's >= 70 % holds for
ordered query sets: extra.general_sorted (78.2 %), extra.general_uniformq (the set XI_j = j/(NQ-1): 77.7 %).

Every line -- any N, --backend group, --config 3 -- carries cpu_baseline (the CPU oracle on this box's host cores: the whole
1e8-query set at N = 1, a 2e7-query sample by rank 0 at N > 1), and at N > 1 extra.n1_reference_ms: rank 0 repeating the
same step alone, so that the 1 -> N ratio can be formed from one line.

Rank 0 prints ONE JSON line (the last line of stdout).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NG = 1_000_000
NQ = 100_000_000
SEED_Q = 0x5EED0003
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SWEEP_MIN_TILES_PER_CU = int(os.environ.get("MI_SWEEP_MIN_TILES_PER_CU", "2"))   # csrc/mi_interp1.hip kSweepMinTilesPerCu


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nq", type=int, default=NQ, help="queries per GPU (default: BASELINE size 1e8)")
    ap.add_argument("--ng", type=int, default=NG)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements (sorted queries, "
                    "implicit-uniform table, config 3, restrict+mean, all-gather)")
    ap.add_argument("--table", choices=["general", "uniform", "nonuniform"], default="general",
                    help="general: explicit X of configs[1] (detected closed form); uniform: implicit grid; nonuniform: "
                         "BASELINE.md section 2's jittered grid X_i = (i + 0.5 u_i)/NG, explicit {x,y} table")
    ap.add_argument("--queries", choices=["random", "sorted", "uniform"], default="random")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI, the real thing) or gloo "
                    "(rehearsal of the N>1 code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--rehearse-one-device", action="store_true",
                    help="map every rank to cuda:0 (only with --dist-backend gloo; timings are then meaningless)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the contract's line): every rank its own --nq queries; strong: ONE set of --nq "
                         "queries split into contiguous NQ/P shards (BASELINE.md section 2), compute-only and compute + "
                         "all-gather both reported")
    ap.add_argument("--shard-of", type=int, default=0, metavar="P",
                    help="1-GPU rehearsal of the strong-scaling shard sizes: time this GPU on shard 0 of P (NQ/P queries)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (process group, barriers, all-reduce of the timings, all-gather) with a "
                         "world of ONE rank: rehearsal of the RCCL calls on a one-GPU box")
    ap.add_argument("--backend", choices=["ranks", "group"], default="ranks",
                    help="ranks (default): one process per GPU, torch.distributed over RCCL; group: one process, the C ABI's "
                         "mi_group_create(N) + mi_group_interp1_f64_dev on device-resident shards")
    ap.add_argument("--gather", action="store_true",
                    help="--backend group: also reassemble the whole result vector on every GPU inside the timed step "
                         "(gathered_dev of mi_group_interp1_f64_dev: an RCCL all-gather over xGMI behind the kernels)")
    ap.add_argument("--gather-chunks", type=int, default=1,
                    help="--backend group --gather: cut every shard into this many chunks and exchange chunk k while the kernel "
                         "of chunk k+1 runs (mi_group_set_gather_chunks); 1 = one kernel, then one ncclAllGather")
    ap.add_argument("--config", type=int, choices=[2, 3], default=2,
                    help="2 (default): BASELINE configs[1], the 1-D headline; 3: configs[2], 4096^2 bilinear, 1e8 scattered "
                         "queries as the timed workload (for profiling interp2_kernel; same JSON contract)")
    return ap.parse_args()


def timed_loop(ctx, fn, steps, warmup, barrier):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize; returns
    (host wall seconds, HIP-event seconds over the same K launches)."""
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    tm = ctx.timer()
    t0 = time.perf_counter()
    tm.start()
    for _ in range(steps):
        fn()
    tm.stop()
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    return wall, tm.elapsed_ms() * 1e-3


def cpu_baseline(X, Y, nq_total, sample=100_000_000, queries="random"):
    """The CPU oracle (kind 'port': the repo's restatement of arma::interp1 semantics, bracket formulation,
    OpenMP over queries) on a bounded sample of the same workload, on this box's host cores.  At N = 1 the sample is the
    whole headline query set; rank 0 of an N > 1 run (whose other ranks wait at a barrier meanwhile) takes 2e7 queries."""
    import numpy as np

    import oracle
    # a 1-GPU box's CPU share is 16 threads (more are visible but belong to other tenants)
    threads = min(oracle.max_threads(), os.cpu_count() or 1, 16)
    n = min(nq_total, sample)
    xi = oracle.splitmix_uniform(SEED_Q, n)
    if queries == "sorted":
        xi = np.sort(xi)
    elif queries == "uniform":
        xi = np.arange(n, dtype=np.float64) / max(n - 1, 1)
    oracle.interp1_bracket(X, Y, xi[:1_000_000], nthreads=threads)      # warm caches / thread pool
    best = None
    for _ in range(3):
        t = time.perf_counter()
        oracle.interp1_bracket(X, Y, xi, nthreads=threads)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    # the literal single-threaded arma::interp1 algorithm (sort XI, resumed scan, un-permute) on a smaller sample
    m = min(n, 20_000_000)
    t = time.perf_counter()
    oracle.interp1_arma(X, Y, xi[:m])
    t_arma = time.perf_counter() - t
    return {"value": n / best, "unit": "points/s", "cores": threads, "kind": "port",
            "sample": "first %d of the %d %s queries (SplitMix64 seed 0x5EED0003), same %d-node table, "
                      "oracle.interp1_bracket (OpenMP), best of 3, %.2f s per pass" % (n, nq_total, queries, len(X), best),
            "arma_interp1_semantics_1thread_points_per_s": m / t_arma,
            "arma_interp1_semantics_sample": "oracle.interp1_arma (sort + resumed scan + un-permute, the literal "
                                             "Armadillo algorithm) on the first %d queries, %.2f s" % (m, t_arma)}


def cpu_baseline_config3(n3, nq_total, sample=20_000_000):
    """CPU oracle (kind 'port') for BASELINE configs[2]: oracle.interp2_bilinear_uniform (OpenMP) on the first `sample`
    query pairs of the 0x5EED0004 stream over the same 4096 x 4096 table."""
    import numpy as np

    import oracle
    threads = min(oracle.max_threads(), os.cpu_count() or 1, 16)
    n = min(nq_total, sample)
    q = oracle.splitmix_uniform(0x5EED0004, 2 * n)
    g = np.arange(n3, dtype=np.float64) / (n3 - 1)
    z = np.sin(2 * np.pi * g)[:, None] * np.cos(2 * np.pi * g)[None, :] + g[None, :] * g[:, None]     # Z(i,j), SURVEY 8d config 3
    h = 1.0 / (n3 - 1)
    oracle.interp2_bilinear_uniform(0.0, h, n3, 0.0, h, n3, z, q[:100_000], q[n:n + 100_000], nthreads=threads)
    best = None
    for _ in range(2):
        t = time.perf_counter()
        oracle.interp2_bilinear_uniform(0.0, h, n3, 0.0, h, n3, z, q[:n], q[n:], nthreads=threads)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return {"value": n / best, "unit": "points/s", "cores": threads, "kind": "port",
            "sample": "first %d of the %d scattered query pairs (SplitMix64 seed 0x5EED0004), same %d x %d table, "
                      "oracle.interp2_bilinear_uniform (OpenMP), best of 2, %.2f s per pass" % (n, nq_total, n3, n3, best)}


def measured_traffic(mode, queries, nq):
    """HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected separately by
    scripts/profile_bench.sh and corrected as MI355X_MICROARCH.md section HBM prescribes); None when the
    committed profile does not cover this kernel/configuration."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        from armadillocudalinearinterpolation_amd import _build
        t = json.load(open(path))
        e = t.get("interp1_mode%d_%s" % (mode, queries))
        if e and int(e.get("nq", 0)) == int(nq):
            if e.get("source_sha256") != _build.source_hash("interp1"):
                return None, "profiles/traffic_latest.json was measured on another build of the kernels (source hash differs): not quoted", {}
            l2 = {k: e[k] for k in ("tcc_req_per_launch", "gpu_clock_hz", "l2_request_bound_ms", "tcc_busy_frac",
                                    "request_cap_per_xcd_per_ns", "request_cap_source") if k in e}
            return e["hbm_bytes_per_launch"], "rocprofv3 PMC passes of this build (scripts/profile_bench.sh), profiles/traffic_latest.json", l2
    except (OSError, ValueError, KeyError):
        pass
    return None, "no committed PMC profile covers this configuration", {}


def measured_traffic_config3(nq):
    """fabric bytes per interp2_kernel launch from the committed PMC profile, quoted only for the same interp2 sources"""
    try:
        from armadillocudalinearinterpolation_amd import _build
        e = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json"))).get("interp2_random")
        if e and int(e.get("nq", 0)) == int(nq) and e.get("source_sha256") == _build.source_hash("interp2"):
            return e["hbm_bytes_per_launch"], "rocprofv3 PMC passes of this build (scripts/profile_bench.sh ... --config 3), profiles/traffic_latest.json"
    except (OSError, ValueError, KeyError):
        pass
    return None, "no committed PMC profile covers this configuration / build"


def kernel_label(args, ginfo, nq, info):
    """which kernel launch_mode (csrc/mi_interp1.hip) picks for this call: region sweep for unordered queries over a table
    beyond L2 with at least SWEEP_MIN_TILES_PER_CU tiles per CU (pipelined form from 16 tiles per CU), else streaming"""
    tiles_per_cu = nq // 16384 / float(info["compute_units"])
    if args.queries == "random" and ginfo["table_bytes"] >= (5 << 20) and tiles_per_cu >= SWEEP_MIN_TILES_PER_CU:
        pipe = tiles_per_cu >= 16 and os.environ.get("MI_SWEEP_VARIANT", "2") == "2"
        return ("interp1_sweep_pipe_kernel<%d,...> (region sweep, pipelined form)" if pipe
                else "interp1_sweep_kernel<%d,...> (region sweep)") % ginfo["mode"]
    return "interp1_vec_kernel<%d,...> (streaming)" % ginfo["mode"]


def bench_config3(args, ctx, info, dev, world, rank, barrier, dist, dist_on):
    """BASELINE configs[2] as the timed workload: 4096 x 4096 fp64 table (arma::mat layout), 1e8 scattered (x, y) queries per
    rank (SplitMix64 seed 0x5EED0004), one step = one mi_interp2_f64_dev pass.  Same JSON contract as the headline."""
    import torch

    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import synth
    n3, nq = 4096, args.nq
    compact = bool(int(os.environ.get("MI_BENCH_GRID2_COMPACT", "0")))
    g2 = mi.Grid2.uniform(ctx, 0.0, 1.0 / (n3 - 1), n3, 0.0, 1.0 / (n3 - 1), n3, synth.config3_table(n3, dev), compact=compact)
    strong = args.scaling == "strong" or args.shard_of > 0
    if strong:      # ONE set of args.nq query pairs (x = stream[0:NQ], y = stream[NQ:2NQ]), contiguous shard per rank
        from armadillocudalinearinterpolation_amd import sharding
        lo, hi = sharding.shard_bounds(args.nq, 0 if args.shard_of > 0 else rank, args.shard_of or world)
        lo -= lo & 1
        nq = hi - lo
        q2 = torch.cat([synth.splitmix_uniform(0x5EED0004, nq, dev, offset=lo),
                        synth.splitmix_uniform(0x5EED0004, nq, dev, offset=args.nq + lo)])
    else:
        q2 = synth.splitmix_uniform(0x5EED0004 + 0x1000 * rank, 2 * nq, dev)
    zq = torch.empty(nq, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    wall, ev = timed_loop(ctx, lambda: g2.interp(q2[:nq], q2[nq:], out=zq), args.steps, args.warmup, barrier)
    t = torch.tensor([wall, ev], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t[0])
    alg = 24.0 * nq + 8.0 * n3 * n3                       # 16 B of queries + 8 B result, table once
    ks = ev / args.steps
    if rank == 0:
        print(json.dumps({
            "metric": "interpolated points/sec (fp64)",
            "value": (args.nq if (strong and not args.shard_of) else world * nq) * args.steps / wall_max, "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "2D bilinear interp, 4096x4096 grid, %.0e scattered query points (BASELINE configs[2])" % nq,
                       "queries_per_gpu": nq, "table_layout": "column pairs (2x input bytes)" if compact else "quad cells (4x input bytes)",
                       "entry_point": "mi_interp2_f64_dev",
                       "path": "interp2_kernel, one random cell per query"},
            "roofline": {"bound": "hbm", "achieved": alg / ks / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": alg / ks / 1e9 / HBM_PEAK_GBPS, "traffic": measured_traffic_config3(nq)[0],
                         "traffic_note": measured_traffic_config3(nq)[1],
                         "kernel": "interp2_kernel",
                         "kernel_ms": ks * 1e3, "algorithmic_bytes_per_launch": alg},
            "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline_config3(n3, nq),
            "device": info["name"]}), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(args):
    """--gpus N > 1 invoked without a launcher: start `torch.distributed.run` (one rank per GPU) as a child process, relay
    what it prints (rank 0's JSON line stays the last line of stdout) and return its exit code.  This parent imports
    neither torch nor the package and never touches the GPU; nothing is exec'ed."""
    import socket
    import subprocess
    with socket.socket() as sk:                       # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            line = ln.rstrip("\n")                    # held back so that it is the LAST line whatever the ranks print after it
        else:
            sys.stdout.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc


def bench_group(args):
    """One host process, N GPUs, through the C ABI only (include/mi355_interp.h "several GPUs of one node"):
    mi_group_create(N), the table replicated by mi_group_grid1_create, one step = ONE mi_group_interp1_f64_dev call over
    N device-resident shards of --nq queries each (every shard's kernel on its own device and stream, concurrently).
    Timed region: mi_group_synchronize on both sides (= every device idle), K steps in between; value = N * nq * K / wall."""
    import torch

    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import synth
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the hot path has no CPU fallback", file=sys.stderr)
        return 2
    n = args.gpus
    devices = [0] * n if args.rehearse_one_device else list(range(n))
    grp = mi.Group(devices)
    rccl_ranks = grp.rccl_ranks()                     # forms the ncclCommInitAll communicator (0: repeated devices)
    if args.gather_chunks > 1:
        grp.set_gather_chunks(args.gather_chunks)
    X, Y = synth.config_grid(args.ng)
    tab = grp.grid1(X, Y, sanitise=False)
    nq = args.nq - (args.nq & 1)
    xq = [synth.splitmix_uniform(SEED_Q + 0x1000 * r, nq, torch.device("cuda", devices[r])) for r in range(n)]
    yq = [torch.empty_like(x) for x in xq]
    full = [torch.empty(n * nq, dtype=torch.float64, device=x.device) for x in xq] if args.gather else None
    for d in set(devices):
        torch.cuda.synchronize(d)
    step = lambda: tab.interp_dev(xq, out=yq, gather=args.gather, gathered=full, sync=False)  # noqa: E731
    for _ in range(args.warmup):
        step()
    grp.synchronize()
    timers = [grp.ctx(r).timer() for r in range(n)]
    t0 = time.perf_counter()
    for tm in timers:
        tm.start()
    for _ in range(args.steps):
        step()
    for tm in timers:
        tm.stop()
    grp.synchronize()
    wall = time.perf_counter() - t0
    ev = max(tm.elapsed_ms() for tm in timers) * 1e-3
    for tm in timers:
        tm.close()
    info = grp.ctx(0).device_info()
    table_bytes = 8.0 * (args.ng + 1)
    alg = 16.0 * nq + table_bytes
    ks = ev / args.steps
    print(json.dumps({
        "metric": "interpolated points/sec (fp64)", "value": n * nq * args.steps / wall, "unit": "points/s",
        "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "backend": "group", "rccl_ranks": rccl_ranks,
        "config": {"workload": "1D linear interp, %.0e random queries per GPU on %.0e-point grid, fp64 (BASELINE configs[1])" % (nq, args.ng),
                   "queries_per_gpu": nq, "grid_nodes": args.ng, "table": "general",
                   "entry_point": "mi_group_interp1_f64_dev" + ((" + gathered_dev (%d chunks, grouped ncclBroadcast behind the kernels)" % args.gather_chunks
                                                                 if args.gather_chunks > 1 else " + gathered_dev (ncclAllGather)") if args.gather else ""),
                   "sharding": "queries/%d, table replicated by mi_group_grid1_create, %s" % (
                       n, "RCCL all-gather of the result shards in the step" if args.gather else "no collective"),
                   "devices": devices},
        "roofline": {"bound": "hbm", "achieved": alg / ks / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": alg / ks / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": "per device: the kernel mi_interp1_f64_dev picks for its shard",
                     "kernel_ms": ks * 1e3, "algorithmic_bytes_per_launch": alg,
                     "note": "slowest device's HIP-event time per step (events on each member's own stream)"},
        "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline(X, Y, nq, sample=100_000_000 if n == 1 else 20_000_000),
        "device": info["name"]}), flush=True)
    tab.close()
    grp.close()
    return 0


def main():
    args = parse()
    if args.backend == "group":
        sys.exit(bench_group(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.force_dist:
        sys.exit(self_launch(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import synth

    if args.force_dist:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29611")):
            os.environ.setdefault(k, v)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist_on = world > 1 or args.force_dist
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print("bench.py --gpus %d inside a world of %d ranks: launch one rank per GPU" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    if args.rehearse_one_device:
        assert args.dist_backend == "gloo", "--rehearse-one-device needs --dist-backend gloo"
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)

        def barrier():
            dist.barrier()
    else:
        def barrier():
            pass

    ctx = mi.Context(local_rank)              # follows torch's current stream on this device
    info = ctx.device_info()
    X, Y = synth.config_grid(args.ng)
    if args.table == "general":
        grid = mi.Grid1.from_nodes(ctx, X, Y, sanitise=False)
    elif args.table == "nonuniform":
        un = synth.splitmix_uniform(0x5EED0002, args.ng, torch.device("cpu")).numpy()
        X = (np.arange(args.ng) + 0.5 * un) / args.ng
        grid = mi.Grid1.from_nodes(ctx, X, Y, sanitise=False)
    else:
        grid = mi.Grid1.uniform(ctx, 0.0, 1.0 / (args.ng - 1), Y)
    ginfo = grid.info()
    nq = args.nq
    if args.config == 3:
        return bench_config3(args, ctx, info, dev, world, rank, barrier, dist, dist_on)
    strong = args.scaling == "strong" or args.shard_of > 0
    if strong:
        # ONE query set of args.nq elements (the headline's SplitMix64 stream), contiguous shard per rank
        from armadillocudalinearinterpolation_amd import sharding
        parts = args.shard_of if args.shard_of > 0 else world
        lo, hi = sharding.shard_bounds(args.nq, 0 if args.shard_of > 0 else rank, parts)
        lo -= lo & 1                                                      # 16-B aligned shard starts (vector kernels)
        nq = hi - lo
        xq = synth.splitmix_uniform(SEED_Q, nq, dev, offset=lo)
    else:
        # each rank's shard of the N*nq-query batch: SplitMix64 stream offset by rank
        xq = synth.splitmix_uniform(SEED_Q + 0x1000 * rank, nq, dev)
    if args.queries == "sorted":
        xq = torch.sort(xq).values
    elif args.queries == "uniform":
        xq = torch.arange(nq, dtype=torch.float64, device=dev) / (nq - 1)
    yq = torch.empty_like(xq)
    torch.cuda.synchronize()

    wall, ev = timed_loop(ctx, lambda: grid.interp(xq, out=yq), args.steps, args.warmup, barrier)
    t = torch.tensor([wall, ev], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max, ev_max = float(t[0]), float(t[1])

    # 8 B query in + 8 B result out per query, plus the resident table read once per launch
    alg_bytes = 16.0 * nq + float(ginfo["table_bytes"])                              # per launch, per GPU
    kernel_s = ev / args.steps                                                        # this rank's avg launch
    achieved = alg_bytes / kernel_s / 1e9
    traffic, traffic_note, l2 = measured_traffic(ginfo["mode"], args.queries, nq)
    result = {
        "metric": "interpolated points/sec (fp64)",
        "value": (args.nq if (strong and not args.shard_of) else world * nq) * args.steps / wall_max,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall_max / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "1D linear interp, %.0e %s queries on %.0e-point grid, fp64 (BASELINE configs[1])"
                        % (nq, args.queries, args.ng),
            "queries_per_gpu": nq, "grid_nodes": args.ng, "table": args.table,
            "table_mode": ginfo["mode"], "table_bytes": ginfo["table_bytes"],
            "entry_point": "mi_interp1_f64_dev", "sharding": "queries/%d, table replicated, no collective" % world,
            "query_set": ("one set of %d queries, contiguous shards of NQ/%d" % (args.nq, args.shard_of or world)) if strong
                         else "every rank its own set of %d queries" % nq,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": traffic_note,
            "kernel": kernel_label(args, ginfo, nq, info),
            "kernel_ms": kernel_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes,
            "note": "achieved = (16 B/query + table bytes) / HIP-event time per launch; traffic: see profiles/",
        },
        "device": info["name"],
    }
    if l2.get("l2_request_bound_ms"):
        # the resource that binds the random-query kernel: every query is one request to the XCD L2s,
        # which serve at most one request per channel per clock.  Requests per launch from the stamped PMC profile of this
        # build; the fraction is against THIS run's kernel time.
        result["roofline"]["l2_request_bound_ms"] = l2["l2_request_bound_ms"]
        result["roofline"]["l2_request_frac"] = l2["l2_request_bound_ms"] / (kernel_s * 1e3)
        result["roofline"]["l2_requests_per_launch"] = l2["tcc_req_per_launch"]
        # the same requests at the rate the XCDs' vector request paths were MEASURED to sustain with all CUs gathering from an
        # L2-resident table (the stamped profile names the figure and the log it comes from): the floor of this algorithm
        if l2.get("request_cap_per_xcd_per_ns"):
            result["roofline"]["request_floor_ms_at_measured_cap"] = l2["tcc_req_per_launch"] / (8 * l2["request_cap_per_xcd_per_ns"] * 1e9) * 1e3
            result["roofline"]["request_cap_source"] = l2.get("request_cap_source")
        result["roofline"]["l2_note"] = ("TCC_REQ per launch / (128 L2 channels x %.2f GHz measured clock): the time the XCD L2s need "
                                         "for the launch's requests at one per channel-clock; profiled TCC_BUSY fraction %.2f"
                                         % (l2["gpu_clock_hz"] * 1e-9, l2.get("tcc_busy_frac") or float("nan")))

    extra = {}
    if rank == 0 and not args.no_extra:
        reps = max(3, min(args.steps, 10))

        def quick(fn):
            # Steady state for the secondary measurements: after any idle of the device (table construction, host work) the
            # first ~8 launches run fast, the next ~50 up to 35 % slower, and only after ~20 ms of continuous launches does
            # the time per call settle (profiles/r04_order_probe_trace.log: 0.25 -> 0.34 -> 0.255 ms for the ordered sets);
            # one call after a change of query set also runs the kernel predicted for the previous set.  So: 30 ms of
            # untimed calls first, then `reps` timed ones.  (The headline above keeps the contract's W warm-up steps.)
            _, e0 = timed_loop(ctx, fn, 3, 2, lambda: None)
            warm = min(400, max(3, int(0.03 / max(e0 / 3, 1e-5))))
            _, e = timed_loop(ctx, fn, reps, warm, lambda: None)
            return e / reps

        xs = torch.sort(xq).values
        xu = torch.arange(nq, dtype=torch.float64, device=dev) / max(nq - 1, 1)          # the uniform set XI_j = j/(NQ-1)
        gu = mi.Grid1.uniform(ctx, 0.0, 1.0 / (args.ng - 1), Y)
        for name, g, q in (("general_sorted", grid, xs), ("general_uniformq", grid, xu), ("uniform_random", gu, xq),
                           ("uniform_sorted", gu, xs)):
            s = quick(lambda: g.interp(q, out=yq))
            extra[name] = {"ms": s * 1e3, "points_per_s": nq / s, "frac_of_8TBps": 16.0 * nq / s / 1e9 / HBM_PEAK_GBPS}
        del xu
        # BASELINE.md section 2's non-uniform variant: X_i = (i + 0.5 u_i)/NG (seed 0x5EED0002), explicit {x,y} table
        un = synth.splitmix_uniform(0x5EED0002, args.ng, torch.device("cpu")).numpy()
        Xn = (np.arange(args.ng) + 0.5 * un) / args.ng
        gn = mi.Grid1.from_nodes(ctx, Xn, Y, sanitise=False)
        for name, q in (("nonuniform_grid_random", xq), ("nonuniform_grid_sorted", xs)):
            s = quick(lambda: gn.interp(q, out=yq))
            extra[name] = {"ms": s * 1e3, "points_per_s": nq / s, "table_mode": gn.info()["mode"],
                           "frac_of_8TBps": (16.0 * nq + gn.info()["table_bytes"]) / s / 1e9 / HBM_PEAK_GBPS}
        del xs, gn
        s = quick(lambda: yq.copy_(xq))
        extra["torch_copy_same_bytes"] = {"ms": s * 1e3, "GBps": 16.0 * nq / s / 1e9}
        # config 3: 4096^2 table, 1e8 scattered queries
        n3 = 4096
        g2 = mi.Grid2.uniform(ctx, 0.0, 1.0 / (n3 - 1), n3, 0.0, 1.0 / (n3 - 1), n3, synth.config3_table(n3, dev))
        q2 = synth.splitmix_uniform(0x5EED0004, 2 * nq, dev)
        s = quick(lambda: g2.interp(q2[:nq], q2[nq:], out=yq))
        b3 = 24.0 * nq + 8.0 * n3 * n3
        extra["config3_bilinear_4096sq"] = {"ms": s * 1e3, "points_per_s": nq / s, "frac_of_8TBps": b3 / s / 1e9 / HBM_PEAK_GBPS}
        del q2, g2
        # config 4's interpolation step: Restrict + masked mean over S=3 x R=1e6
        S, R = 3, 1_000_000
        t0 = torch.rand(S * R, device=dev) * 5
        t1 = torch.rand(S * R, device=dev) + 5
        i0 = torch.randint(0, 1000, (S * R,), device=dev, dtype=torch.int16)
        i1 = i0 + 1
        acc = torch.ones(R, device=dev, dtype=torch.int32)
        s = quick(lambda: mi.restrict_mean(ctx, t0, i0, t1, i1, acc, 5.0, 3.0, 1024, S))
        extra["restrict_mean_3x1e6"] = {"ms": s * 1e3, "elements_per_s": S * R / s,
                                         "GBps": (12.0 * S * R + 4.0 * R) / s / 1e9}
        del t0, t1, i0, i1, acc
        # config 4's residual evaluation at one GPU's share: 125 000 realisations x 1024 grid points
        for mode, name in ((mi.MATH_EXACT, "exact"), (mi.MATH_FAST, "fast")):
            edm = mi.EventDrivenMap(ctx, [13.0589], 125_000, n_grid=1024, math_mode=mode)
            zd = [0.3310, 0.6914, 1.3557]
            edm.ComputeF(zd)
            edm.ComputeF(zd)
            tmg = edm.last_timings()
            extra["compute_f_125k_real_1024pts_%s" % name] = {
                "ms": tmg["total_ms"], "evolve_ms": tmg["evolve_ms"], "restrict_mean_ms": tmg["restrict_mean_ms"],
                "realisations_per_s": 125_000 / (tmg["total_ms"] * 1e-3),
                "note": "compute-bound in Evolve (fp32 exp/log/div), not a bandwidth roofline case"}
            edm.close()
        # the same share on the grid of the reference's Driver.cu (N = 512, configs[4])
        edm = mi.EventDrivenMap(ctx, [13.0589], 125_000, n_grid=512, math_mode=mi.MATH_EXACT)
        edm.ComputeF(zd)
        edm.ComputeF(zd)
        tmg = edm.last_timings()
        extra["compute_f_125k_real_512pts_exact"] = {
            "ms": tmg["total_ms"], "evolve_ms": tmg["evolve_ms"], "restrict_mean_ms": tmg["restrict_mean_ms"],
            "realisations_per_s": 125_000 / (tmg["total_ms"] * 1e-3)}
        edm.close()
        # the same call with the opt-in shortcut for sigma = 0 (config 4 has sigma = 0: its realisations are R copies of
        # one computation); reported beside the full evolution above, never instead of it
        edm = mi.EventDrivenMap(ctx, [13.0589], 125_000, n_grid=1024, dedup_identical=1)
        edm.ComputeF(zd)
        edm.ComputeF(zd)
        tmg = edm.last_timings()
        extra["compute_f_125k_real_1024pts_exact_dedup_identical"] = {
            "ms": tmg["total_ms"], "evolve_ms": tmg["evolve_ms"], "restrict_mean_ms": tmg["restrict_mean_ms"],
            "note": "opt-in mi_edm_params.dedup_identical: one realisation evolved, events replicated to all rows; "
                    "outputs bit-identical to the full evolution (tests/test_edm_gpu.py); only valid for sigma = 0"}
        edm.close()
    if strong and dist_on and args.dist_backend == "nccl":
        # "also report compute + all-gather": the result shards reassembled on every GPU by an RCCL
        # all-gather over xGMI inside the timed loop (ragged shards are padded to the largest)
        from armadillocudalinearinterpolation_amd import sharding as _sh
        sizes = [b - a for a, b in (_sh.shard_bounds(args.nq, r, world) for r in range(world))]
        kmax = max(sizes) + 1
        pad = torch.zeros(kmax, dtype=torch.float64, device=dev)
        full = torch.empty(world * kmax, dtype=torch.float64, device=dev)

        def step_ag():
            grid.interp(xq, out=pad[:nq])
            dist.all_gather_into_tensor(full, pad)
        wall_ag, _ = timed_loop(ctx, step_ag, args.steps, args.warmup, barrier)
        tt = torch.tensor([wall_ag], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        extra_strong = {"compute_only_ms_per_step": wall_max / args.steps * 1e3,
                        "compute_plus_allgather_ms_per_step": float(tt[0]) / args.steps * 1e3,
                        "points_per_s_compute_only": args.nq * args.steps / wall_max,
                        "points_per_s_compute_plus_allgather": args.nq * args.steps / float(tt[0]),
                        "allgather": "RCCL all_gather_into_tensor of %d x %d B shards over xGMI, in the timed loop" % (world, 8 * kmax)}
        del pad, full
    else:
        extra_strong = None
    if dist_on and not args.no_extra and args.dist_backend == "nccl" and not strong:
        full = torch.empty(world * nq, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(full, yq)
        torch.cuda.synchronize()
        dist.barrier()
        t = time.perf_counter()
        dist.all_gather_into_tensor(full, yq)
        torch.cuda.synchronize()
        dist.barrier()
        extra["allgather_ms"] = (time.perf_counter() - t) * 1e3
        extra["allgather_note"] = "RCCL all-gather of %d x 8e8 B result shards, outside the timed region" % world
        del full
    if dist_on and world > 1:
        # a reader of ONE file should be able to form the 1 -> N ratio: after the timed region rank 0 repeats the very same
        # step alone (the other ranks wait at the barrier below, their GPUs idle) -- the N = 1 time of this shard size
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            w1, e1 = timed_loop(ctx, lambda: grid.interp(xq, out=yq), args.steps, args.warmup, lambda: None)
            extra["n1_reference_ms"] = w1 / args.steps * 1e3
            extra["n1_reference_note"] = ("rank 0 alone, same %d-query shard, %d steps, the other %d ranks idle at a barrier: "
                                          "ms_per_step / n1_reference_ms is the slow-down of one shard when all %d GPUs run"
                                          % (nq, args.steps, world - 1, world))
    if rank == 0:
        # the CPU baseline beside EVERY line (north_star: "reported at 1, 2, 4 and 8 GPUs alongside that CPU baseline"): the
        # whole 1e8-query set at N = 1, a 2e7-query sample at N > 1 (the other ranks wait at the closing barrier meanwhile)
        result["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(
            X, Y, nq, sample=100_000_000 if world == 1 else 20_000_000, queries=args.queries)
        result["extra"] = extra
        if extra_strong:
            result["strong_scaling"] = extra_strong
        print(json.dumps(result), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
