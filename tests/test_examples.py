"""The example scripts run end to end on the GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script", ["iris_mala_single_chain.py", "iris_hmc_multichain.py",
                                    "iris_diagnostics_and_prediction.py", "deep_narrow_hmc.py"])
def test_example_runs(script):
    env = dict(os.environ, EEYORE_EXAMPLE_EPOCHS="33", EEYORE_EXAMPLE_CHAINS="96", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "cceptance rate" in out.stdout


@pytest.mark.parametrize("ladder", [1, 8])
def test_config5_example_runs_on_one_gpu(ladder):
    """examples/mnist_shaped_tempering.py (BASELINE config 5 through the sampler surface) as one process: plain HMC on the
    layerwise path (world = 1), and with EEYORE_EXAMPLE_LADDER=8 the whole 8-temperature ladder in this process with
    label exchanges every five iterations."""
    env = dict(os.environ, EEYORE_EXAMPLE_EPOCHS="10", EEYORE_EXAMPLE_CHAINS="12", EEYORE_EXAMPLE_ROWS="128",
               EEYORE_EXAMPLE_LADDER=str(ladder), PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "mnist_shaped_tempering.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "kernel family: bgemm" in out.stdout and f"{ladder} temperature(s) x 12 chains" in out.stdout
    assert "Iterations per chain: 10" in out.stdout and "cceptance rate" in out.stdout
    if ladder > 1:
        swaps = int(out.stdout.rsplit("label exchanges accepted:", 1)[1].split()[0])
        assert swaps > 0


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
    # --steps is rounded up to whole launches of 25 iterations (what HMC.run issues), at least five of them (one 21 ms launch is
    # one sample of a quantity that moves by 10 %); `steps` is what was timed and the line says what was asked for
    assert d["n_gpus"] == 1 and d["steps"] == 125 and d["config"]["steps_requested"] == 3 and d["vs_baseline"] is None
    assert d["config"]["iterations_per_launch"] == 25 and "rounded up" in d["config"]["steps_note"]
    assert d["config"]["launches_timed"] == 5
    assert 0 < d["config"]["launch_ms_min"] <= d["config"]["launch_ms_median"] <= d["config"]["launch_ms_max"]
    assert abs(d["ms_per_step"] * 125 - 5 * d["config"]["launch_ms_median"]) < 0.5 * d["ms_per_step"] * 125
    # the timed region records the chains as HMC.run does, and the same workload through the sampler surface is beside it
    assert d["config"]["recorded_in_timed_region"].startswith("samples [steps, C, P]")
    via = d["config"]["through_sampler_run"]
    assert via["iterations"] == 125 and 0.5 < via["ratio_to_headline"] < 1.5 and 0 < via["acceptance"] <= 1
    assert 0 < d["roofline"]["frac_exact_f32_products"] < d["roofline"]["frac"] and 0 < d["roofline"]["bf16_pipe_flops_frac"] < 1
    # the arithmetic type is f32; the default form of the kernel's products is named beside it, both forms' rates recorded
    assert d["dtype"] == "f32 (bf16x3 products, f32 accumulate)" and d["config"]["f32_products"] == "bf16x3"
    assert set(d["config"]["kernels"]) == {"bf16x3", "exact"} and d["roofline"]["f32_equivalent"] is True
    assert d["config"]["kernel"] == "mfma32" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] == "TFLOP/s"
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["value"] > 100 * cb["value"]
    # the reference's own op sequence timed for BASELINE configs[0] too, the device's figure for it beside it
    c1 = cb["reference_faithful_config1"]
    assert "error" not in c1 and c1["value"] > 0 and c1["gpu_same_config"]["value"] > c1["value"]
    # the ESS gather runs on the driver-sized run too, and the secondary figure (config 5's share of one GPU) is beside the metric
    assert d["config"]["ess"]["num_chains"] == 512 and d["config"]["ess"]["mean"] > 0
    sec = d["config"]["secondary"]["config5_share_one_gpu"]
    assert "error" not in sec and sec["kernel"] == "bgemm" and 0.2 < sec["frac_of_f32_mfma_peak"] < 1.5 and 0 <= sec["acceptance"] <= 1
