"""PowerPosteriorSampler: ladder, partner distribution and swap rule against the reference's vectors (G6), and the
run-loop / exchange bookkeeping on CPU with the oracle test double."""
import numpy as np
import pytest
import torch
from torch.distributions import Normal
from torch.utils.data import DataLoader

from eeyore_amd.constants import loss_functions
from eeyore_amd.datasets import XYDataset
from eeyore_amd.models import mlp
from eeyore_amd.samplers import PowerPosteriorSampler
from tests.helpers import load
from tests.oracle_plan import attach


def _setup(R=1, K=5, between_step=1, device="cpu", seed=5):
    z = load("g6_power_posterior.npz")
    dt = torch.float64
    hp = mlp.Hyperparameters(dims=[2, 3, 2, 1], bias=3 * [True], activations=3 * [torch.sigmoid])
    m = mlp.MLP(loss=loss_functions['binary_classification'], hparams=hp, dtype=dt, device=device)
    m.prior = Normal(torch.tensor(z["prior_mu"], dtype=dt, device=device), torch.tensor(z["prior_sigma"], dtype=dt, device=device))
    if device == "cpu":
        attach(m, vector_temp=True)
    ds = XYDataset(torch.tensor(z["x"], dtype=dt, device=device), torch.tensor(z["y"], dtype=dt, device=device))
    loader = DataLoader(ds, batch_size=len(ds), shuffle=False)
    torch.manual_seed(seed)
    th0 = 0.5 * torch.randn(20, dtype=dt) if R == 1 else 0.5 * torch.randn(R, 20, dtype=dt)
    s = PowerPosteriorSampler(m, loader, [['MALA', {'step': 0.1}] for _ in range(K)], theta0=th0.to(device),
                              between_step=between_step, b=0.5, rng='torch')
    return z, m, ds, s


def test_ladder_and_partner_distribution_match_reference():
    z, m, ds, s = _setup()
    np.testing.assert_allclose(s.temperature, z["ladder"], rtol=1e-15)  # t_i = (i/K)^4, power_posterior_sampler.py:92
    for i in range(5):
        np.testing.assert_allclose(s.eval_categorical_probs(i), z[f"cat_probs/{i}"], rtol=1e-12)
        assert abs(s.eval_categorical_probs(i).sum() - 1) < 1e-12
    for (i, j), lq in zip(z["pairs"], z["log_q"]):
        np.testing.assert_allclose(s.categorical_log_prob(i, j).item(), lq[0], rtol=1e-6)  # torch Categorical is f32
        np.testing.assert_allclose(s.categorical_log_prob(j, i).item(), lq[1], rtol=1e-6)


def test_swap_log_rate_matches_reference():
    z, m, ds, s = _setup()
    K = 5
    # put the reference's recorded states into the ladder
    s.sampler.set_current(torch.tensor(z["samples"]), data=(ds.x, ds.y))
    np.testing.assert_allclose(s._ell()[:, 0].numpy(), z["ell"], rtol=1e-10)
    plan = m._plan(ds.x, ds.y)
    for (i, j), want in zip(z["pairs"], z["log_rate"]):
        args = s.between_chain_move_log_rate(int(i), torch.tensor([int(j)]))
        swap, lr = plan.pt_swap_decide(args[0], args[1], args[2], args[3], torch.tensor([0.5], dtype=torch.float64),
                                       dlogq=args[4])
        np.testing.assert_allclose(lr.item(), want, rtol=1e-5, atol=1e-5)  # log q terms are f32 in the reference


def test_run_exchanges_states_consistently():
    R = 6
    z, m, ds, s = _setup(R=R)
    s.run(num_epochs=25, num_burnin_epochs=5)
    K = s.num_chains
    for k in range(K):
        assert s.get_chain(k).get_samples().shape == (20, R, 20)
    assert s.get_chain() is s.chains[K - 1]  # default indicator: the t = 1 chain
    # after all the exchanges every chain's cached tempered target/gradient equals a fresh evaluation at its state
    plan = m._plan(ds.x, ds.y)
    t, g = plan.log_target_grad(s.sampler._theta.clone(), temp=s._tvec)
    np.testing.assert_allclose(s.sampler._target.numpy(), t.numpy(), rtol=1e-9)
    np.testing.assert_allclose(s.sampler._grad.numpy(), g.numpy(), rtol=1e-8, atol=1e-10)
    n_swaps = sum(int(sw.sum()) for _, sw, _ in s.last_swaps)
    assert 0 <= n_swaps <= K * R


def test_exchange_is_a_permutation_of_states():
    R = 4
    z, m, ds, s = _setup(R=R, between_step=1000)
    for _ in range(3):
        s.within_chain_moves(ds.x, ds.y)  # the ladder starts from one state per replica: let the chains separate
    before = s.sampler._theta.clone()
    torch.manual_seed(3)
    s._rand = lambda n: torch.full((n,), 1e-12, dtype=torch.float64)  # log u very negative: accept every exchange
    s.between_chain_moves(ds.x, ds.y)
    after = s.sampler._theta
    for r in range(R):
        a = sorted(map(tuple, before[r::R].numpy().round(12).tolist()))
        b = sorted(map(tuple, after[r::R].numpy().round(12).tolist()))
        assert a == b  # states move between temperatures within a replica, never across replicas
    assert not torch.equal(before, after)


@pytest.mark.gpu
def test_power_posterior_on_gpu_with_hip_decide_kernel():
    R = 64
    z, m, ds, s = _setup(R=R, device="cuda:0")
    s.sampler.rng = 'philox'
    s.run(num_epochs=40, num_burnin_epochs=10)
    plan = m._plan(ds.x, ds.y)
    t, g = plan.log_target_grad(s.sampler._theta.clone(), temp=s._tvec)
    np.testing.assert_allclose(s.sampler._target.cpu().numpy(), t.cpu().numpy(), rtol=1e-9)
    np.testing.assert_allclose(s.sampler._grad.cpu().numpy(), g.cpu().numpy(), rtol=1e-8, atol=1e-10)
    cold = s.get_chain().get_target_vals().mean().item()
    hot = s.get_chain(0).get_target_vals().mean().item()
    assert np.isfinite(cold) and np.isfinite(hot)
    assert s.get_chain().get_samples().shape == (30, R, 20)


def test_multi_chain_surface_and_file_storage(tmp_path):
    """The reference's multi-chain surface (multi_chain_serial_sampler.py:10-46) and storage='file'
    (power_posterior_sampler.py:57-66): get_param / get_sample / reset_chains / to_chainfile / set_current / set_all, the
    chain<i> directories a file-backed sampler appends to, and what they hold against the in-memory run of the same seed."""
    from eeyore_amd.chains import ChainFile
    z, m, ds, s = _setup(R=1, between_step=2)
    torch.manual_seed(11)
    s.run(num_epochs=9, num_burnin_epochs=3)
    K, P = s.num_chains, 20
    assert s.get_param(4).shape == (6,) and s.get_param(4, chain_idx=0).shape == (6,)
    assert torch.equal(s.get_param(4), s.get_chain().get_samples()[:, 0, 4])
    assert s.get_sample(2).shape == (P,) and torch.equal(s.get_sample(2, chain_idx=1), s.get_chain(1).get_samples()[2, 0])
    s.to_chainfile(path=tmp_path / "dump", mode='w')
    for i in range(K):   # the reference's folder names: 'sampler' + str(i).zfill(num_chains)
        back = ChainFile(keys=['sample', 'target_val'], path=tmp_path / "dump" / ('sampler' + str(i).zfill(K)), mode='a')
        cl = back.to_chainlist()
        np.testing.assert_array_equal(torch.stack(cl.vals['sample']).numpy(), s.get_chain(i).get_samples()[:, 0].numpy())
        np.testing.assert_array_equal(torch.stack(cl.vals['target_val']).numpy(), s.get_chain(i).get_target_vals()[:, 0].numpy())
    mem = [s.get_chain(i).get_samples()[:, 0].clone() for i in range(K)]
    s.reset_chains()
    assert all(len(s.get_chain(i)) == 0 for i in range(K))
    # set_current / set_all: every temperature at the given state, evaluated there
    th = torch.linspace(-0.3, 0.3, P, dtype=torch.float64)
    for setter in (s.set_current, s.set_all):
        setter(th)
        assert torch.equal(s.sampler._theta, th.expand(K, P))
        plan = m._plan(ds.x, ds.y)
        t, _ = plan.log_target_grad(s.sampler._theta.clone(), temp=s._tvec)
        np.testing.assert_allclose(s.sampler._target.numpy(), t.numpy(), rtol=1e-12)
    # the same run, stored on file: chain<i+1>/<key>.csv, appended iteration by iteration
    z2, m2, ds2, f = _setup(R=1, between_step=2)
    f2 = PowerPosteriorSampler(m2, f.dataloader, [['MALA', {'step': 0.1}] for _ in range(K)], theta0=f.sampler._theta[0].clone(),
                               between_step=2, b=0.5, rng='torch', storage='file', path=tmp_path / "run", mode='a')
    torch.manual_seed(11)
    f2.run(num_epochs=9, num_burnin_epochs=3)
    for i in range(K):
        folder = tmp_path / "run" / f"chain{i + 1}"
        assert sorted(p.name for p in folder.iterdir()) == ['sample.csv', 'target_val.csv']
        got = ChainFile(keys=['sample', 'target_val'], path=folder, mode='a').to_chainlist()
        np.testing.assert_array_equal(torch.stack(got.vals['sample']).numpy(), mem[i].numpy())   # '%.18e' round-trips f64
    with pytest.raises(RuntimeError, match="on file"):
        f2.get_param(0)
    with pytest.raises(ValueError):
        PowerPosteriorSampler(m2, f.dataloader, [['MALA', {'step': 0.1}]] * K, theta0=th, storage='tape')


def test_file_storage_with_several_replicas(tmp_path):
    z, m, ds, s0 = _setup(R=3)
    K = s0.num_chains
    s = PowerPosteriorSampler(m, s0.dataloader, [['MALA', {'step': 0.1}] for _ in range(K)], theta0=s0.sampler._theta[:3].clone(),
                              between_step=1, b=0.5, rng='torch', storage='file', keys=['sample', 'target_val', 'accepted'],
                              path=tmp_path, mode='a')   # ('w' would truncate at every update, as the reference's ChainFile does)
    s.run(num_epochs=5, num_burnin_epochs=1)
    from eeyore_amd.chains import ChainFile
    for i in (0, K - 1):
        for r in range(3):
            folder = tmp_path / f"chain{i + 1}" / f"replica{r + 1}"
            cl = ChainFile(keys=['sample', 'target_val', 'accepted'], path=folder, mode='a').to_chainlist()
            assert len(cl.vals['sample']) == 4 and cl.vals['sample'][0].shape == (20,) and set(cl.vals['accepted']) <= {0, 1}
