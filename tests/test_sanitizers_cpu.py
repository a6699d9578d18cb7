"""ASan + UBSan over the CPU-side code (oracle C restatements, C++ host layer): GPU sanitizers are not
available on this pool, so memory/UB checking happens on the CPU builds."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_oracle_and_host_layer_are_sanitizer_clean():
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "sanitize_cpu.sh")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sanitize_cpu: clean" in out.stdout
