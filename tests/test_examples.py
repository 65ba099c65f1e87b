"""The drop-in example scripts (counterparts of the reference's examples/samplers/mlp/iris) run end to end."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script", ["mala_gpu_chainlist.py", "hmc_gpu_multichain.py"])
def test_example_runs(script):
    env = dict(os.environ, EEYORE_EXAMPLE_EPOCHS="33", EEYORE_EXAMPLE_CHAINS="96", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "samplers", "mlp", "iris", script)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "cceptance rate" in out.stdout
