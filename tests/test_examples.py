"""The example scripts run end to end on the GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script", ["iris_mala_single_chain.py", "iris_hmc_multichain.py"])
def test_example_runs(script):
    env = dict(os.environ, EEYORE_EXAMPLE_EPOCHS="33", EEYORE_EXAMPLE_CHAINS="96", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "cceptance rate" in out.stdout
