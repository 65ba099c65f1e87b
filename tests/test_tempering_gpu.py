"""BASELINE config 5's algorithm on the GPU: the label-exchange sweep on the device (HIP swap decision,
device Philox) against K ranks simulated on the host; PowerPosteriorSampler with HMC at eight temperatures on the layerwise
path (MLP(784-128-10)): swap log-rates against the oracle, cached target / gradient after exchanges and relabelling against
re-evaluation.  (eeyore/samplers/power_posterior_sampler.py:128-182.)"""
import numpy as np
import pytest
import torch
from torch.distributions import Normal
from torch.utils.data import DataLoader

from oracle import mlp_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _torch_decide(ell_i, ell_j, t_i, t_j, u, dlogq=None):
    lr = (t_i - t_j) * (ell_j - ell_i)
    return (torch.log(u) < lr).to(torch.uint8), lr


def _bus_rank(bus):
    from eeyore_amd.distributed import TemperingExchange

    class BusRank(TemperingExchange):
        """A rank whose all-gather reads what every simulated rank posted for this sweep."""
        def _gather(self, t):
            return torch.stack(bus["ell" if t.dtype.is_floating_point else "labels"])
    return BusRank


def test_label_exchange_on_the_device_equals_eight_ranks_on_the_host():
    """K = 8 ladder positions x R replicas: ``LocalTemperingLadder`` on the GPU (all pairs of a sweep through one
    ey_pt_swap_decide launch, accept variates from ey_philox_uniform) against eight ``TemperingExchange`` ranks on the
    host that all-gather through a bus (torch decision rule, variates from the host entry point ey_philox_block): ladder
    positions after every sweep, swap counts and temperature vectors equal."""
    from eeyore_amd.distributed import LocalTemperingLadder
    K, R = 8, 257
    temps = [(i / K) ** 4 for i in range(1, K + 1)]
    ladder = LocalTemperingLadder(temps, R, DEV, seed=11)
    bus = {}
    ranks = [_bus_rank(bus)(temps, R, g, K, "cpu", seed=11, decide=_torch_decide) for g in range(K)]
    total = 0
    for sweep in range(12):
        ell = [(-60.0 + 15.0 * torch.randn(R, generator=torch.Generator().manual_seed(100 * sweep + g), dtype=torch.float64))
               for g in range(K)]
        bus["ell"], bus["labels"] = ell, [r.labels.clone() for r in ranks]
        counts = [int(r.exchange(ell[r.rank])) for r in ranks]
        assert len(set(counts)) == 1                                   # every rank took the same decisions
        got = int(ladder.exchange(torch.stack(ell).to(DEV).reshape(-1)))
        host = torch.stack([r.labels for r in ranks])
        assert torch.equal(ladder.labels.cpu(), host), sweep
        assert got == counts[0]
        assert torch.equal(host.sort(0).values, torch.arange(K)[:, None].expand(K, R))   # a permutation per replica
        total += got
    assert ladder.num_swaps == ranks[0].num_swaps == total > 0
    tv = ladder.temperature_vector(torch.float32).cpu().view(K, R)
    for g in range(K):
        assert torch.equal(tv[g], ranks[g].temperature_vector(torch.float32))


def _mnist_shaped(rows, dtype=torch.float32):
    rng = np.random.default_rng(0)
    x = (rng.random((rows, 784)) * (rng.random((rows, 784)) < 0.19)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[np.arange(rows) % 10]
    return torch.tensor(x, dtype=dtype, device=DEV), torch.tensor(y, dtype=dtype, device=DEV)


def _mlp_784(dtype=torch.float32):
    from eeyore_amd.constants import loss_functions
    from eeyore_amd.models import mlp
    model = mlp.MLP(loss=loss_functions['multiclass_classification'],
                    hparams=mlp.Hyperparameters(dims=[784, 128, 10], activations=[torch.sigmoid, None]), dtype=dtype, device=DEV)
    P = model.num_params()
    model.prior = Normal(torch.zeros(P, device=DEV), torch.ones(P, device=DEV))
    return model, P


def test_power_posterior_hmc_at_eight_temperatures_on_the_layerwise_path():
    """PowerPosteriorSampler([['HMC', ...]] * 8) on MLP(784-128-10): the 8 x R chains are one chain batch with a per-chain
    temperature vector on the layerwise ("bgemm") path.  After every iteration (within-chain HMC, then the reference's
    sequential between-chain moves) the recorded swap log-rates equal the oracle's formula on the recorded inputs, and the
    cached tempered log-target and gradient of every chain -- rescaled, not re-evaluated, when states moved -- equal a
    fresh evaluation at the chain's temperature."""
    from eeyore_amd.datasets import XYDataset
    from eeyore_amd.samplers import PowerPosteriorSampler
    K, R, rows = 8, 3, 96
    model, P = _mlp_784()
    x, y = _mnist_shaped(rows)
    loader = DataLoader(XYDataset(x, y), batch_size=rows, shuffle=False)
    torch.manual_seed(2)
    s = PowerPosteriorSampler(model, loader, [['HMC', {'step': 0.002, 'num_steps': 3}] for _ in range(K)],
                              theta0=0.05 * torch.randn(R, P, device=DEV), between_step=1, b=0.5, rng='philox', seed=3)
    plan = model._plan(x, y)
    assert plan.kernel == "bgemm" and s.sampler._theta.shape == (K * R, P)
    temps = np.array(s.temperature)
    log_q = s._log_q.double().cpu().numpy()
    moved = 0
    for it in range(4):
        before = s.sampler._theta.clone()
        s.draw(x, y, savestate=True)
        s.counter.increment_idx()
        for i, ((j, swap, log_rate), (ell_i, ell_j, t_i, t_j, dlogq)) in enumerate(zip(s.last_swaps, s.last_swap_inputs)):
            jn = j.cpu().numpy()
            want = mlp_oracle.pt_swap_log_rate(log_q[jn, i], log_q[i, jn], ell_i.double().cpu().numpy(),
                                               ell_j.double().cpu().numpy(), temps[i], temps[jn])
            np.testing.assert_allclose(log_rate.double().cpu().numpy(), want, rtol=2e-4, atol=2e-3)
            moved += int(swap.sum())
        t2, g2 = plan.log_target_grad(s.sampler._theta.clone(), temp=s._tvec)
        np.testing.assert_allclose(s.sampler._target.cpu().numpy(), t2.cpu().numpy(), rtol=2e-4, atol=2e-3)
        gs = float(g2.abs().max())
        np.testing.assert_allclose(s.sampler._grad.cpu().numpy(), g2.cpu().numpy(), rtol=2e-3, atol=2e-4 * gs)
        assert not torch.equal(before, s.sampler._theta)
    assert moved > 0                                                    # exchanges did happen
    for k in range(K):
        assert s.get_chain(k).get_samples().shape == (4, R, P)
    assert s.get_param(7).shape == (4, R) and s.get_sample(1, chain_idx=0).shape == (R, P)


def test_relabelled_hmc_replicas_keep_their_state_on_the_layerwise_path():
    """The multi-GPU form of the same algorithm on one device: HMC chains on the layerwise path whose temperature vector
    CHANGES between blocks (``LocalTemperingLadder`` + ``set_temperature``, as examples/mnist_shaped_tempering.py runs it):
    after every relabelling the cached tempered target / gradient equal a fresh evaluation at the new temperatures, and the
    next block starts from them."""
    from eeyore_amd.datasets import XYDataset
    from eeyore_amd.distributed import LocalTemperingLadder
    from eeyore_amd.samplers import HMC
    K, R, rows = 8, 4, 64
    model, P = _mlp_784()
    x, y = _mnist_shaped(rows)
    loader = DataLoader(XYDataset(x, y), batch_size=rows, shuffle=False)
    ladder = LocalTemperingLadder([(i / K) ** 4 for i in range(1, K + 1)], R, DEV, seed=5)
    torch.manual_seed(4)
    sampler = HMC(model, theta0=0.05 * torch.randn(K * R, P, device=DEV), dataloader=loader, step=0.002, num_steps=3, seed=9,
                  temperature=ladder.temperature_vector(torch.float32))
    plan = model._plan(x, y)
    assert plan.kernel == "bgemm"
    swaps = 0
    for block in range(4):
        sampler.run(num_epochs=2, num_burnin_epochs=0)
        swaps += int(ladder.exchange(sampler.current['target_val'] / sampler.temperature))
        new_t = ladder.temperature_vector(torch.float32)
        sampler.set_temperature(new_t)
        t2, g2 = plan.log_target_grad(sampler._theta.clone(), temp=new_t)
        np.testing.assert_allclose(sampler._target.cpu().numpy(), t2.cpu().numpy(), rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(sampler._grad.cpu().numpy(), g2.cpu().numpy(), rtol=2e-3, atol=2e-4 * float(g2.abs().max()))
    assert swaps > 0 and not torch.equal(ladder.labels, torch.arange(K, device=DEV)[:, None].expand(K, R))
