"""N > 1 path on the GPU box: two ranks (gloo, both on cuda:0 -- the box has one GPU) run scripts/run_newton.py's sharded
ComputeF / Newton loop; the result must equal the single-rank run.  With sigma = 0 the partial sums are exact
((R-1) x in fp64), so the sharded residual and every Newton iterate are BIT-identical to the unsharded ones, in the
reference's averaging (default) and with the true mean; with sigma > 0 the fp64 sums are added in another order and
the residual agrees to 1 ulp(fp32) of the mean."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "run_newton.py")


def _run(world, extra, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world == 1:
        cmd = [sys.executable, SCRIPT] + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port), SCRIPT, "--backend", "gloo", "--one-device"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mean", [[], ["--true-mean"]])
def test_two_rank_rehearsal_equals_single_rank(mean):
    port = 29700 + os.getpid() % 1000
    # residual, sigma = 0: bit-identical; odd R so the shards differ in size (1001 / 1000)
    a = ["--real", "2001", "--threads", "512", "--residual-only"] + mean
    one, two = _run(1, a, port), _run(2, a, port + 1)
    assert one["mean"] == ("true" if mean else "reference") and two["n_gpus"] == 2 and two["realisations_per_gpu"] == 1001
    assert one["f"] == two["f"]
    # residual, sigma = 0.3: realisations differ, the shards' sums add up in another order
    a = ["--real", "3000", "--threads", "512", "--residual-only", "--sigma", "0.3"] + mean
    one, two = _run(1, a, port + 2), _run(2, a, port + 3)
    assert np.allclose(one["f"], two["f"], rtol=0, atol=2e-7) and np.all(np.isfinite(one["f"]))
    # the replicated Newton loop over the sharded residual (config 5's structure): identical iterates
    a = ["--real", "1200", "--threads", "1024", "--max-iterations", "3"] + mean
    one, two = _run(1, a, port + 4), _run(2, a, port + 5)
    assert one["iterations"] == two["iterations"] and one["iterations"] >= 1
    assert one["solution"] == two["solution"] and one["history"] == two["history"]
