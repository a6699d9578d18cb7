"""Short runs of the differential fuzzers in scripts/ (seeded, a few seconds each) so that the GPU suite itself walks
random grid families, sizes, query styles, order hints and model parameters against the oracle.  The long runs of the
round are quoted in DESIGN.md section 5."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

SCRIPTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(SCRIPTS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_interp1_fuzz_short(mi_ctx):
    res = _load("gpu_fuzz_interp1").run(12.0, 2024, ctx=mi_ctx)
    assert res["cases"] >= 50 and set(res["modes"]) >= {0, 2, 3}


def test_interp2_fuzz_short(mi_ctx):
    res = _load("gpu_fuzz_interp2").run(8.0, 2024, ctx=mi_ctx)
    assert res["cases"] >= 100 and res["scalar"] >= 10


def test_edm_fuzz_short(mi_ctx, monkeypatch):
    monkeypatch.delenv("MI_EDM_WAVES_PER_REALISATION", raising=False)
    res = _load("gpu_fuzz_edm").run(10.0, 2024, ctx=mi_ctx)
    assert res["cases"] >= 30 and res["filled"] >= 3         # (filled: launches of 3200 realisations, more than the device holds at once)
