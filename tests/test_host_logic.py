"""Host-side logic on CPU: C-ABI surface, model/param layout, counters, chains, sampler run loop.

The sampler tests replace the HIP plan with tests/oracle_plan.OraclePlan (a test double on the C oracle) so that the
reference's run-loop semantics can be checked without a GPU; the recorded randomness of the golden traces is fed
through the samplers' rng hooks, so the whole trace must reproduce."""
import ctypes as ct
import os
import re

import numpy as np
import pytest
import torch
from torch.distributions import Normal
from torch.utils.data import DataLoader

from eeyore_amd import _lib as L
from eeyore_amd.chains import ChainBuffer, ChainFile, ChainList, ChainLists
from eeyore_amd.constants import loss_functions
from eeyore_amd.datasets import DataCounter, XYDataset, synthetic
from eeyore_amd.models import mlp
from eeyore_amd.samplers import HMC, MALA, MetropolisHastings
from tests.helpers import groups, load
from tests.oracle_plan import attach

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ACTS = {0: None, 1: torch.sigmoid, 2: torch.tanh, 3: torch.relu}
LIKS = {0: 'binary_classification', 1: 'multiclass_classification'}


# ------------------------------------------------------------------------------------------------ C ABI surface
def test_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "eeyore_amd.h")).read()
    declared = set(re.findall(r"\b(ey_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    lib = L.lib()  # raises if the .so is missing
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ey_version() >= 100


def test_plan_refuses_cpu_device_loudly():
    from eeyore_amd.plan import Plan
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Plan([2, 2, 1], [1, 1], [1, 1], 0, torch.float64, "cpu")


def test_argument_errors_without_gpu():
    lib = L.lib()
    h = ct.c_void_p()
    dims = (ct.c_int * 3)(2, 2, 1)
    ones = (ct.c_int * 2)(1, 1)
    assert lib.ey_plan_create(ct.byref(h), 0, dims, ones, ones, 0, 0, 0) == -1
    assert b"n_layers" in lib.ey_last_error()
    assert lib.ey_plan_create(ct.byref(h), 2, dims, ones, ones, 7, 0, 0) == -1
    assert lib.ey_plan_create(ct.byref(h), 2, dims, ones, (ct.c_int * 2)(9, 1), 0, 0, 0) == -1
    assert lib.ey_plan_destroy(None) == 0


# ------------------------------------------------------------------------------------------------ model surface
def test_hyperparameters_validation():
    with pytest.raises(ValueError):
        mlp.Hyperparameters(dims=[2, 1], bias=[True], activations=[torch.sigmoid])  # mlp.py:15-16
    with pytest.raises(ValueError):
        mlp.Hyperparameters(dims=[2, 2, 1], bias=[True, True], activations=[torch.sigmoid])  # mlp.py:18-19


def test_param_layout_matches_reference_tests():
    # tests/test_binary_classif_mlp2321_log_lik.py:50-64 of the reference: W_l row-major then b_l, layer by layer
    hp = mlp.Hyperparameters(dims=[2, 3, 2, 1], bias=3 * [True], activations=3 * [torch.sigmoid])
    m = mlp.MLP(loss=loss_functions['binary_classification'], hparams=hp)
    assert m.num_params() == 20
    th = torch.arange(20, dtype=torch.float64)
    m.set_params(th.clone())
    assert torch.equal(m.fc_layers[0].weight, th[0:6].view(3, 2)) and torch.equal(m.fc_layers[0].bias, th[6:9])
    assert torch.equal(m.fc_layers[1].weight, th[9:15].view(2, 3)) and torch.equal(m.fc_layers[1].bias, th[15:17])
    assert torch.equal(m.fc_layers[2].weight, th[17:19].view(1, 2)) and torch.equal(m.fc_layers[2].bias, th[19:20])
    assert torch.equal(m.get_params(), th)


def test_model_rejects_unknown_loss_and_activation():
    hp = mlp.Hyperparameters(dims=[2, 2, 1], activations=[torch.sigmoid, torch.nn.functional.gelu])
    m = mlp.MLP(loss=loss_functions['binary_classification'], hparams=hp)
    with pytest.raises(ValueError, match="no HIP kernel"):
        m._plan(torch.zeros(4, 2), torch.zeros(4, 1))
    m2 = mlp.MLP(loss=lambda x, y: (x - y).sum(), hparams=mlp.Hyperparameters(dims=[2, 2, 1]))
    with pytest.raises(ValueError, match="loss_functions"):
        m2._plan(torch.zeros(4, 2), torch.zeros(4, 1))


def test_loss_callables_match_reference_formulas():
    x = torch.tensor([[0.2], [0.9]], dtype=torch.float64)
    y = torch.tensor([[0.], [1.]], dtype=torch.float64)
    assert loss_functions['binary_classification'](x, y).item() == pytest.approx(-(np.log(0.8) + np.log(0.9)))
    lg = torch.tensor([[1., 2., 0.5]], dtype=torch.float64)
    oh = torch.tensor([[0., 1., 0.]], dtype=torch.float64)
    want = -(2.0 - np.log(np.exp(1) + np.exp(2) + np.exp(0.5)))
    assert loss_functions['multiclass_classification'](lg, oh).item() == pytest.approx(want)


# ------------------------------------------------------------------------------------------------ datasets / counters
def test_bundled_datasets_equal_reference_data():
    z = load("datasets.npz")
    xor = XYDataset.from_eeyore('xor', dtype=torch.float64)
    iris = XYDataset.from_eeyore('iris', yndmin=1, yonehot=True, dtype=torch.float64)
    np.testing.assert_array_equal(xor.x.numpy(), z["xor_x"]); np.testing.assert_array_equal(xor.y.numpy(), z["xor_y"])
    np.testing.assert_array_equal(iris.x.numpy(), z["iris_x"]); np.testing.assert_array_equal(iris.y.numpy(), z["iris_y"])


def test_synthetic_iris_shaped_matches_fixture_generator():
    rec = groups(load("g4_hmc_traces.npz"))["mlp432323_synth"]
    x, y = synthetic.iris_shaped_arrays(seed=0)
    np.testing.assert_array_equal(x, rec["x"]); np.testing.assert_array_equal(y, rec["y"])
    assert x.shape == (150, 4) and y.sum(0).tolist() == [50, 50, 50]


def test_xydataset_batched_fetch_yields_the_batches_of_the_row_by_row_fetch():
    """XYDataset.__getitems__ (one gather per batch) hands the DataLoader the rows, order and values that one
    __getitem__ per row does (eeyore/datasets/xydataset.py:21-22), for shuffled full batches and for minibatches, and
    consumes the global generator identically."""
    from torch.utils.data import DataLoader
    from eeyore_amd.datasets import XYDataset

    class RowByRow(XYDataset):
        __getitems__ = None  # the fetcher then falls back to __getitem__

    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(37, 4, generator=g, dtype=torch.float64), torch.randn(37, 3, generator=g, dtype=torch.float64)
    for batch_size in (37, 10):
        for yy in (y, y[:, 0]):
            torch.manual_seed(11)
            a = [(bx.clone(), by.clone()) for bx, by in DataLoader(XYDataset(x, yy), batch_size=batch_size, shuffle=True)]
            after_a = torch.rand(1)
            torch.manual_seed(11)
            b = [(bx.clone(), by.clone()) for bx, by in DataLoader(RowByRow(x, yy), batch_size=batch_size, shuffle=True)]
            after_b = torch.rand(1)
            assert len(a) == len(b) and torch.equal(after_a, after_b)
            for (ax, ay), (bx, by) in zip(a, b):
                assert ax.shape == bx.shape and ay.shape == by.shape
                assert torch.equal(ax, bx) and torch.equal(ay, by)


def test_batches_equals_iterating_the_dataloader_including_the_random_stream():
    """datasets.batches (what SerialSampler.run walks): the batches of the DataLoader itself and the same state of the
    global generator after every epoch, so a seeded script draws the same momenta / proposals as it would behind the
    loader (eeyore/samplers/serial_sampler.py:44-45 iterates the loader).  Custom collate functions fall through."""
    from torch.utils.data import DataLoader
    from eeyore_amd.datasets import XYDataset, batches
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(23, 4, generator=g, dtype=torch.float64), torch.randn(23, 2, generator=g, dtype=torch.float64)
    for kw in (dict(batch_size=23, shuffle=True), dict(batch_size=23, shuffle=False), dict(batch_size=5, shuffle=True),
               dict(batch_size=5, shuffle=True, drop_last=True), dict(batch_size=7, shuffle=False)):
        for yy in (y, y[:, 0]):
            got, want, states = [], [], []
            for walk, store in ((batches, got), (iter, want)):
                torch.manual_seed(17)
                loader = DataLoader(XYDataset(x, yy), **kw)
                for epoch in range(3):
                    for bx, by in walk(loader):
                        store.append((bx.clone(), by.clone(), torch.randn(2)))  # a draw between batches, as a sampler makes
                states.append(torch.get_rng_state())
            assert len(got) == len(want) and torch.equal(states[0], states[1]), kw
            for a, b in zip(got, want):
                assert a[0].shape == b[0].shape and a[1].shape == b[1].shape
                assert all(torch.equal(u, v) for u, v in zip(a, b)), kw
    own = DataLoader(XYDataset(x, y), batch_size=4, collate_fn=lambda rows: (len(rows), rows[0][0]))
    assert [b[0] for b in batches(own)] == [4, 4, 4, 4, 4, 3]


def test_data_counter():
    c = DataCounter(batch_size=50, sample_size=150)
    assert c.num_batches == 3
    c = DataCounter(batch_size=64, sample_size=150)
    assert c.num_batches == 3
    c = DataCounter(batch_size=64, sample_size=150, drop_last=True)
    assert c.num_batches == 2
    c.set_epoch_info(10, 2)
    assert (c.num_iters, c.num_burnin_iters) == (20, 4)
    c.set_iter_info(7, 3)
    assert (c.num_epochs, c.num_burnin_epochs) == (4, 2)
    c.increment_idx(); c.increment_idx(3)
    assert c.idx == 4
    c.reset()
    assert c.idx == 0
    ds = XYDataset(torch.zeros(10, 2), torch.zeros(10, 1))
    c = DataCounter.from_dataloader(DataLoader(ds, batch_size=10))
    assert c.num_batches == 1


# ------------------------------------------------------------------------------------------------ chains
def test_chain_list_and_file_round_trip(tmp_path):
    ch = ChainList()
    for i in range(5):
        ch.detach_and_update(dict(sample=torch.full((3,), float(i), dtype=torch.float64),
                                  target_val=torch.tensor(-float(i), dtype=torch.float64), accepted=i % 2, extra=1))
    assert len(ch) == 5 and ch.num_params() == 3
    assert ch.acceptance_rate() == 2 / 5
    assert torch.equal(ch.get_param(1), torch.arange(5, dtype=torch.float64))
    assert torch.equal(ch.mean(), torch.full((3,), 2.0, dtype=torch.float64))
    ch.to_chainfile(path=tmp_path, mode='w')
    line = open(tmp_path / "sample.csv").readline().strip()
    assert line == ",".join(["0.000000000000000000e+00"] * 3)  # '%.18e' rows (chain_file.py:28-45)
    back = ChainFile(keys=['sample', 'target_val', 'accepted'], path=tmp_path, mode='a').to_chainlist()
    assert torch.equal(back.get_samples(), ch.get_samples()) and back.vals['accepted'] == ch.vals['accepted']
    both = ChainLists.from_chain_list([ch, back])
    assert both.num_chains() == 2 and both.num_samples() == 5 and both.acceptance() == [0.4, 0.4]
    ch.save(tmp_path / "c.pt")
    ch2 = ChainList(); ch2.load(tmp_path / "c.pt")
    assert torch.equal(ch2.get_samples(), ch.get_samples())


def test_chain_buffer_views():
    buf = ChainBuffer()
    C, P = 3, 4
    for i in range(7):
        buf.detach_and_update(dict(sample=torch.full((C, P), float(i)) + torch.arange(C)[:, None],
                                   target_val=torch.full((C,), -float(i)),
                                   accepted=torch.tensor([1, 0, i % 2], dtype=torch.uint8), grad_val=None))
    assert len(buf) == 7 and buf.num_chains() == C and buf.num_params() == P
    assert buf.get_samples().shape == (7, C, P)
    np.testing.assert_allclose(buf.acceptance_rate().numpy(), [1.0, 0.0, 3 / 7])
    c2 = buf.get_chain(2)
    assert isinstance(c2, ChainList) and len(c2) == 7 and c2.acceptance_rate() == 3 / 7
    assert torch.equal(c2.get_sample(3), torch.full((P,), 5.0))
    assert buf.to_chainlists().num_chains() == C


# ------------------------------------------------------------------------------------------------ samplers
def _model_from(rec, dtype=torch.float64):
    dims = rec["dims"].tolist()
    hp = mlp.Hyperparameters(dims=dims, bias=[True] * (len(dims) - 1), activations=[ACTS[a] for a in rec["acts"]])
    m = mlp.MLP(loss=loss_functions[LIKS[int(rec["lik"])]], hparams=hp, dtype=dtype)
    m.prior = Normal(torch.tensor(rec["prior_mu"], dtype=dtype), torch.tensor(rec["prior_sigma"], dtype=dtype))
    return attach(m)


def _feed(sampler, rec):
    """Route the sampler's torch-RNG hooks to the recorded streams of a golden trace."""
    it = {"i": 0}
    sampler._randn = lambda C, P: torch.tensor(rec["z"][it["i"]])[None]
    def rand(C):
        v = torch.tensor([rec["u"][it["i"]]])
        it["i"] += 1
        return v
    sampler._rand = rand


@pytest.mark.parametrize("key,kind", [("cfg1_big_step", "hmc"), ("mlp2321", "hmc"), ("mala_mlp2321", "mala"),
                                      ("mh_mlp2321", "mh")])
def test_sampler_run_reproduces_reference_trace(key, kind):
    rec = groups(load("g4_hmc_traces.npz" if kind == "hmc" else "g5_mala_mh_traces.npz"))[key]
    m = _model_from(rec)
    ds = XYDataset(torch.tensor(rec["x"]), torch.tensor(rec["y"]))
    loader = DataLoader(ds, batch_size=len(ds), shuffle=False)
    th0 = torch.tensor(rec["theta0"])
    if kind == "hmc":
        s = HMC(m, theta0=th0, dataloader=loader, step=float(rec["step"]), num_steps=int(rec["L"]), chain=ChainList())
    elif kind == "mala":
        s = MALA(m, theta0=th0, dataloader=loader, step=float(rec["par"]), chain=ChainList())
    else:
        s = MetropolisHastings(m, theta0=th0, dataloader=loader, chain=ChainList())
        s.kernel.set_density_params(th0.clone(), scale=torch.full_like(th0, float(rec["par"])))
    np.testing.assert_allclose(s.current['target_val'].item(), rec["init_target"], rtol=1e-12)
    _feed(s, rec)
    iters, burn = rec["z"].shape[0], 5
    s.run(num_epochs=iters, num_burnin_epochs=burn)
    ch = s.get_chain()
    assert len(ch) == iters - burn  # serial_sampler.py:46: savestate only once idx >= num_burnin_iters
    assert ch.vals['accepted'] == rec["accepted"][burn:].tolist()
    np.testing.assert_allclose(ch.get_samples().numpy(), rec["sample"][burn:], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(ch.get_target_vals().numpy(), rec["target_val"][burn:], rtol=1e-9)
    assert ch.acceptance_rate() == pytest.approx(rec["accepted"][burn:].mean())
    assert torch.equal(m.get_params(), s.current['sample'])  # the model's parameters follow the chain


def test_batched_sampler_stores_into_chain_buffer():
    rec = groups(load("g5_mala_mh_traces.npz"))["mala_mlp2321"]
    m = _model_from(rec)
    ds = XYDataset(torch.tensor(rec["x"]), torch.tensor(rec["y"]))
    loader = DataLoader(ds, batch_size=len(ds), shuffle=False)
    C = 6
    torch.manual_seed(0)
    th0 = 0.3 * torch.randn(C, 20, dtype=torch.float64)
    s = MALA(m, theta0=th0, dataloader=loader, step=0.3, rng='torch')
    assert s.batched and isinstance(s.chain, ChainBuffer)
    s.run(num_epochs=12, num_burnin_epochs=2)
    assert s.chain.get_samples().shape == (10, C, 20)
    acc = s.chain.acceptance_rate()
    assert acc.shape == (C,) and 0 < acc.mean() < 1
    assert s.current['accepted'].dtype == torch.uint8 and s.current['accepted'].shape == (C,)


def test_hmc_leapfrog_surface():
    rec = groups(load("g3_leapfrog.npz"))["f64/mlp2321/e0.1_L10"]
    m = _model_from(rec)
    x, y = torch.tensor(rec["x"]), torch.tensor(rec["y"])
    loader = DataLoader(XYDataset(x, y), batch_size=4)
    s = HMC(m, theta0=torch.tensor(rec["theta0"]), dataloader=loader, step=0.1, num_steps=10)
    th, p, t, g = s.leapfrog(torch.tensor(rec["theta0"]), torch.tensor(rec["p0"]), x, y)
    np.testing.assert_allclose(th.numpy(), rec["thetaL"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(p.numpy(), rec["pL"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(t.item(), rec["target"], rtol=1e-10)


def test_dual_averaging_tuner_matches_reference_recurrence():
    """HMCDATuner against a direct evaluation of Hoffman & Gelman's algorithm 5 recurrences with the reference's
    constants (eeyore/tuners/hmcda_tuner.py:43-59)."""
    from eeyore_amd.tuners import HMCDATuner
    tuner = HMCDATuner(l=1.0, e0=0.2, d=0.65, eub=0.5)
    barh, logbare, m = 0.0, 0.0, np.log(10 * 0.2)
    rates = [0.9, 0.2, 0.75, 1.0, 0.4, 0.66, 0.1]
    for idx, rate in enumerate(rates):
        it = idx + 1
        d_w, e_w = 1 / (it + 10), 1 / it ** 0.75
        barh = (1 - d_w) * barh + d_w * (0.65 - rate)
        loge = min(m - np.sqrt(it) * barh / 0.05, np.log(0.5))
        logbare = e_w * loge + (1 - e_w) * logbare
        last = idx == len(rates) - 1
        e, n = tuner.tune(rate, idx, return_e=not last)
        want = np.exp(logbare if last else loge)
        assert e == pytest.approx(want, rel=1e-12) and n == max(1, round(1.0 / want))
    assert HMCDATuner(l=2.0).m is None and HMCDATuner(l=2.0).num_steps(0.3) == 7


def test_hmc_with_dual_averaging_tuner_adapts_step():
    from eeyore_amd.tuners import HMCDATuner
    rec = groups(load("g4_hmc_traces.npz"))["mlp2321"]
    m = _model_from(rec)
    ds = XYDataset(torch.tensor(rec["x"]), torch.tensor(rec["y"]))
    loader = DataLoader(ds, batch_size=len(ds), shuffle=False)
    torch.manual_seed(4)
    s = HMC(m, theta0=torch.tensor(rec["theta0"]), dataloader=loader, tuner=HMCDATuner(l=2.0, eub=2.0), chain=ChainList())
    assert 0 < s.step <= 2.0 and s.num_steps == max(1, round(2.0 / s.step))  # init_step heuristic (hmc.py:38-77)
    first = s.step
    s.run(num_epochs=60, num_burnin_epochs=40)
    assert s.step != first and s.num_steps == max(1, round(2.0 / s.step))
    assert len(s.get_chain()) == 20 and 0.2 < s.get_chain().acceptance_rate() <= 1.0


def test_per_chain_dual_averaging_matches_scalar_tuner_elementwise():
    from eeyore_amd.tuners import HMCDATuner, PerChainDATuner
    e0 = torch.tensor([0.05, 0.2, 0.7], dtype=torch.float64)
    batched = PerChainDATuner(e0, num_steps=7, d=0.65, eub=1.0)
    scalars = [HMCDATuner(l=1.0, e0=float(e), d=0.65, eub=1.0) for e in e0]
    rng = np.random.default_rng(0)
    for idx in range(12):
        rates = rng.random(3)
        last = idx == 11
        got, L = batched.tune(torch.tensor(rates), idx, return_e=not last)
        want = [s.tune(r, idx, return_e=not last)[0] for s, r in zip(scalars, rates)]
        np.testing.assert_allclose(got.numpy(), want, rtol=1e-12)
        assert L == 7


def test_hmc_with_per_chain_tuner_adapts_each_chain():
    from eeyore_amd.tuners import PerChainDATuner
    rec = groups(load("g4_hmc_traces.npz"))["mlp2321"]
    m = _model_from(rec)
    ds = XYDataset(torch.tensor(rec["x"]), torch.tensor(rec["y"]))
    loader = DataLoader(ds, batch_size=len(ds), shuffle=False)
    torch.manual_seed(1)
    C = 5
    th0 = 0.3 * torch.randn(C, 20, dtype=torch.float64)
    e0 = torch.tensor([0.05, 0.1, 0.3, 0.8, 2.0], dtype=torch.float64)
    s = HMC(m, theta0=th0, dataloader=loader, tuner=PerChainDATuner(e0, num_steps=6), rng='torch')
    assert torch.equal(s.step, e0) and s.num_steps == 6
    s.run(num_epochs=50, num_burnin_epochs=40)
    assert s.step.shape == (C,) and not torch.equal(s.step, e0)
    assert s.step.max() / s.step.min() < e0.max() / e0.min()  # very different starts move towards each other
    assert s.get_chain().get_samples().shape == (10, C, 20)
