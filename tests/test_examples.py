"""The example scripts run end to end on the GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script", ["iris_mala_single_chain.py", "iris_hmc_multichain.py",
                                    "iris_diagnostics_and_prediction.py"])
def test_example_runs(script):
    env = dict(os.environ, EEYORE_EXAMPLE_EPOCHS="33", EEYORE_EXAMPLE_CHAINS="96", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "cceptance rate" in out.stdout


def test_bench_contract_small_run():
    """bench.py prints ONE JSON line with the driver's fields, roofline and cpu_baseline objects."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2",
                          "--chains-per-gpu", "512"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["vs_baseline"] is None
    # the arithmetic type is f32; the default form of the kernel's products is named beside it, both forms' rates recorded
    assert d["dtype"] == "f32 (bf16x3 products, f32 accumulate)" and d["config"]["f32_products"] == "bf16x3"
    assert set(d["config"]["kernels"]) == {"bf16x3", "exact"} and d["roofline"]["f32_equivalent"] is True
    assert d["config"]["kernel"] == "mfma32" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] == "TFLOP/s"
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["value"] > 100 * cb["value"]
