"""bench.py's launch shapes: plain `--gpus N` self-launches one rank per GPU as a child torch.distributed.run, and
`--backend group` drives the C ABI's mi_group_* path in one process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--nq", "2000000", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline"]


def _run(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _last_json(out):
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert lines, out.stderr[-2000:]
    return json.loads(lines[-1])


def test_plain_multi_gpu_invocation_starts_a_launcher_child_and_relays_its_exit_code():
    """No GPU here: the two ranks the parent starts must each report that the hot path has no CPU fallback, and the parent
    (which itself never touches torch.cuda) must hand their failure back as its own exit code -- not the old
    'must be launched with torch.distributed.run' refusal."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher relay")
    out = _run("--gpus", "2", "--dist-backend", "gloo", "--rehearse-one-device", *SMALL)
    assert out.returncode != 0
    assert "must be launched with torch.distributed.run" not in out.stderr
    # the ranks ran and said why they stopped (the launcher may terminate the second one before it has printed, once the
    # first has failed: one message is enough to show that the children were started and their failure relayed)
    assert out.stderr.count("bench.py needs a GPU") >= 1


@pytest.mark.gpu
def test_plain_gpus_2_self_launches_two_ranks_and_prints_one_line():
    """`python3 bench.py --gpus 2 ...` with no launcher around it: two rank processes (gloo rendezvous on 127.0.0.1, both
    on cuda:0 -- a rehearsal, timings meaningless) and ONE JSON line, the last line of stdout, with n_gpus = 2."""
    out = _run("--gpus", "2", "--dist-backend", "gloo", "--rehearse-one-device", "--nq", "2000000", "--steps", "2", "--warmup", "1",
               "--no-extra")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = _last_json(out)
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["scaling"] == "weak"
    assert r["value"] > 0 and r["config"]["queries_per_gpu"] == 2_000_000
    assert sum(1 for ln in out.stdout.splitlines() if '"metric"' in ln) == 1
    # every N > 1 line carries the CPU baseline (north_star: "reported at 1, 2, 4 and 8 GPUs alongside that CPU baseline")
    # and the one-rank time of the same shard, so that the 1 -> N ratio can be formed from this one line
    assert r["cpu_baseline"]["value"] > 0 and r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["cores"] >= 1
    assert r["extra"]["n1_reference_ms"] > 0


@pytest.mark.gpu
def test_group_backend_reports_the_rccl_communicator_size():
    """--backend group on the one GPU of the box: mi_group_create(1) + mi_group_interp1_f64_dev; RCCL is bound by dlopen,
    ncclCommInitAll forms a one-rank communicator and ncclCommCount reports it."""
    out = _run("--backend", "group", "--gpus", "1", *SMALL)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = _last_json(out)
    assert r["backend"] == "group" and r["n_gpus"] == 1 and r["rccl_ranks"] == 1
    assert r["config"]["entry_point"].startswith("mi_group_interp1_f64_dev") and r["value"] > 0
    out = _run("--backend", "group", "--gpus", "1", "--nq", "2000000", "--steps", "2", "--warmup", "1")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert _last_json(out)["cpu_baseline"]["value"] > 0                    # the group line carries the CPU baseline too
    # the all-gather form on the same communicator
    out = _run("--backend", "group", "--gpus", "1", "--gather", *SMALL)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = _last_json(out)
    assert r["rccl_ranks"] == 1 and "ncclAllGather" in r["config"]["entry_point"]


@pytest.mark.gpu
def test_group_backend_rehearsal_with_a_repeated_device():
    """three shards, all on GPU 0: the shard arithmetic and the concurrent per-shard streams run; RCCL cannot form a
    communicator over a repeated device, which the line says with rccl_ranks = 0."""
    out = _run("--backend", "group", "--gpus", "3", "--rehearse-one-device", *SMALL)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = _last_json(out)
    assert r["n_gpus"] == 3 and r["rccl_ranks"] == 0 and r["config"]["devices"] == [0, 0, 0]


@pytest.mark.gpu
def test_config3_line_carries_roofline_and_cpu_baseline():
    """--config 3 (BASELINE configs[2]) at a small query count: one line, roofline of interp2_kernel, CPU baseline from the
    bilinear oracle on the same 4096 x 4096 table."""
    out = _run("--config", "3", "--nq", "2000000", "--steps", "2", "--warmup", "1", timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = _last_json(out)
    assert r["roofline"]["kernel"] == "interp2_kernel" and r["roofline"]["frac"] > 0
    assert r["cpu_baseline"]["value"] > 0 and "interp2_bilinear_uniform" in r["cpu_baseline"]["sample"]
