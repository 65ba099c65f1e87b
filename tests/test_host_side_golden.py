"""Host-side pieces around the step against fixtures captured from the reference (G9, tests/golden/make_golden.py):
HMCDATuner.tune (hmcda_tuner.py:43-59), HMC.init_step (hmc.py:38-77), ChainFile's bytes (chain_file.py:21-45),
ChainLists.from_file (chain_lists.py:29-36) and the directory SerialSampler.benchmark leaves (serial_sampler.py:54-126).
The kernel-backed halves of these (init_step and a tuned burn-in through the C ABI) are in test_gpu_parity.py."""
import math
from pathlib import Path

import numpy as np
import torch

from oracle import mlp_oracle as orc
from tests.helpers import load, spec_from, subgroups


def _g9(prefix):
    z = load("g9_host_side.npz")
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def _tuner_cases():
    z = _g9("tuner/")
    for key in sorted({k.split("/")[0] for k in z}):
        l, e0, d, eub = z[f"{key}/args"]
        yield key, float(l), float(e0), float(d), (None if math.isnan(eub) else float(eub)), z[f"{key}/rates"], \
            z[f"{key}/step"], z[f"{key}/num_steps"], z[f"{key}/final_state"]


def test_dual_averaging_tuner_reproduces_reference_sequences():
    from eeyore_amd.tuners import HMCDATuner
    n = 0
    for key, l, e0, d, eub, rates, steps, nsteps, final in _tuner_cases():
        t = HMCDATuner(l, e0=e0, d=d, eub=eub)
        got_e, got_n = [], []
        for i, r in enumerate(rates):
            e, k = t.tune(float(r), i, return_e=i < len(rates) - 1)
            got_e.append(e); got_n.append(k)
        np.testing.assert_allclose(got_e, steps, rtol=1e-12)
        assert got_n == nsteps.tolist()
        np.testing.assert_allclose([t.barh, t.logbare, t.m], final, rtol=1e-12, atol=1e-15)
        # the oracle's restatement
        oe, on, ofinal = orc.dual_averaging(l, e0, d, eub, rates)
        np.testing.assert_allclose(oe, steps, rtol=1e-13)
        assert on.tolist() == nsteps.tolist()
        np.testing.assert_allclose(ofinal, final, rtol=1e-13, atol=1e-16)
        n += 1
    assert n == 3


def test_per_chain_tuner_is_the_scalar_recurrence_elementwise():
    """PerChainDATuner fed the reference's rate sequences in separate lanes must reproduce each
    reference step sequence in its lane (the number of leapfrog steps is fixed there, so only the steps compare)."""
    from eeyore_amd.tuners import PerChainDATuner
    cases = [c for c in _tuner_cases() if c[4] is None]
    e0 = torch.tensor([c[2] for c in cases], dtype=torch.float64)
    t = PerChainDATuner(e0, num_steps=7, d=cases[0][3])
    assert all(c[3] == cases[0][3] for c in cases)
    n = len(cases[0][5])
    for i in range(n):
        rate = torch.tensor([c[5][i] for c in cases], dtype=torch.float64)
        e, k = t.tune(rate, i, return_e=i < n - 1)
        np.testing.assert_allclose(e.numpy(), [c[6][i] for c in cases], rtol=1e-12)
        assert k == 7


def test_oracle_init_step_matches_reference():
    n = 0
    for key, rec in subgroups(load("g9_host_side.npz"), "init_step").items():
        spec = spec_from(rec)
        got = orc.init_step(spec, rec["theta0"], rec["momentum"], rec["x"], rec["y"])
        assert got == float(rec["step"]), key
        assert max(1, round(1.0 / got)) == int(rec["num_steps"])
        np.testing.assert_allclose(math.log(10 * got), float(rec["tuner_m"]), rtol=1e-15)
        n += 1
    assert n == 3


def test_chain_files_are_byte_identical_to_the_reference(tmp_path):
    from eeyore_amd.chains import ChainFile, ChainList, ChainLists
    z = _g9("chainfile/")
    n = len(z["accepted"])
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        ch = ChainList(keys=["sample", "target_val", "grad_val", "accepted"])
        for i in range(n):
            ch.update(dict(sample=torch.tensor(z["sample"][i], dtype=dt), target_val=torch.tensor(z["target_val"][i], dtype=dt),
                           grad_val=torch.tensor(z["grad_val"][i], dtype=dt), accepted=int(z["accepted"][i])))
        d = tmp_path / tag
        ch.to_chainfile(path=d, mode="w")
        for k in ("sample", "target_val", "grad_val", "accepted"):
            assert (d / f"{k}.csv").read_bytes() == z[f"{tag}/{k}.csv"].tobytes(), (tag, k)
        back = ChainFile(keys=["sample", "target_val", "accepted"], path=d).to_chainlist(dtype=dt)
        assert np.array_equal(torch.stack(back.vals["sample"]).numpy(), z[f"{tag}/readback_sample"])
        assert np.array_equal(torch.stack(back.vals["target_val"]).numpy(), z[f"{tag}/readback_target_val"])
        assert back.vals["accepted"] == z[f"{tag}/readback_accepted"].tolist()
        # the reference's own bytes read back through ChainLists.from_file
        ref = tmp_path / f"ref_{tag}"
        ref.mkdir()
        for k in ("sample", "target_val", "accepted"):
            (ref / f"{k}.csv").write_bytes(z[f"{tag}/{k}.csv"].tobytes())
        cl = ChainLists.from_file([ref, d], dtype=dt)
        assert tuple(cl.get_samples().shape) == (2, n, z["sample"].shape[1])
        assert torch.equal(cl.get_samples()[0], cl.get_samples()[1])
    # appending one state per call, the defaults of ChainFile.update (reopen in 'a', write, close)
    d = tmp_path / "append"
    cf = ChainFile(keys=["sample", "target_val", "accepted"], path=d, mode="a")
    cf.close()
    for i in range(3):
        cf.update(dict(sample=torch.tensor(z["sample"][i]), target_val=torch.tensor(z["target_val"][i]),
                       accepted=int(z["accepted"][i])))
    for k in ("sample", "target_val", "accepted"):
        assert (d / f"{k}.csv").read_bytes() == z[f"append3/{k}.csv"].tobytes(), k


def test_benchmark_leaves_the_reference_directory_layout(tmp_path):
    """SerialSampler.benchmark's bookkeeping with a stand-in sampler (no kernel involved): runNN/<key>.csv + runtime.txt
    per successful chain, run_counts.txt, a rejected run and a crashed run counted as the reference counts them."""
    from eeyore_amd.chains import ChainList, ChainLists
    from eeyore_amd.datasets import DataCounter
    from eeyore_amd.samplers.base import SerialSampler
    z = _g9("benchmark/")
    num_chains, num_epochs, burn = (int(v) for v in z["args"])

    class Walk(SerialSampler):
        def __init__(self):
            super().__init__(DataCounter(1, 1))  # one batch per epoch
            self.dataloader = [(None, None)]
            self.chain = ChainList()
            self.calls = 0

        def get_chain(self):
            return self.chain

        def reset(self, theta, data=None, reset_counter=True, reset_chain=True):
            self.counter.reset()
            self.chain.reset(keys=self.chain.vals.keys())
            self.state = theta.clone()
            self.calls += 1
            if self.calls == 2:
                raise RuntimeError("injected failure")  # what benchmark() catches (serial_sampler.py:112)

        def draw(self, x, y, savestate=False):
            self.state = self.state + 1
            if savestate:
                self.chain.update(dict(sample=self.state.clone(), target_val=self.state.sum(), accepted=1))

    s = Walk()
    seen = []

    def conditions(chain, runtime):
        seen.append(len(chain))
        return len(seen) != 1  # the first finished run is turned down

    s.benchmark(num_chains=num_chains, num_epochs=num_epochs, num_burnin_epochs=burn, path=tmp_path,
                init=[torch.zeros(9, dtype=torch.float64)] * 8, check_conditions=conditions)
    listing = sorted(str(p.relative_to(tmp_path)) for p in Path(tmp_path).rglob("*") if "errors" not in str(p))
    assert listing == z["listing"].tolist()
    assert (tmp_path / "run_counts.txt").read_text() == "3,succesful\n1,unmet_conditions\n1,runtime_errors\n"
    assert z["run_counts.txt"].tobytes().decode() == "3,succesful\n0,unmet_conditions\n0,runtime_errors\n"
    lines = [len((tmp_path / "run1" / f"{k}.csv").read_text().splitlines()) for k in ("sample", "target_val", "accepted")]
    assert lines == z["lines_per_file"].tolist() == [num_epochs - burn] * 3
    assert len(list((tmp_path / "run1" / "errors").glob("error*.txt"))) == 1
    cl = ChainLists.from_file([tmp_path / f"run{i}" for i in (1, 2, 3)])
    assert list(cl.get_samples().shape) == z["from_file_shape"].tolist()
    float((tmp_path / "run2" / "runtime.txt").read_text())
