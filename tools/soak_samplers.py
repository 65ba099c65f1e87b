#!/usr/bin/env python3
"""The sampler surface end to end on every kernel family and both dtypes: HMC with a per-chain dual-averaging tuner, MALA and
random-walk MH, chain statistics attached, burn-in then stored iterations.  usage: python tools/soak_samplers.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from torch.distributions import Normal
from torch.utils.data import DataLoader
from eeyore_amd.constants import loss_functions
from eeyore_amd.datasets import XYDataset
from eeyore_amd.models import mlp
from eeyore_amd.samplers import HMC, MALA, MetropolisHastings
from eeyore_amd.tuners import PerChainDATuner
from eeyore_amd.distributed import ChainStats
DEV = 'cuda:0'
rng = np.random.default_rng(0)
for dims, acts, lik, dt in (([4, 16, 3], [torch.sigmoid, None], 'multiclass_classification', torch.float32),
                            ([4, 20, 3], [torch.tanh, None], 'multiclass_classification', torch.float64),
                            ([10, 100, 10], [torch.sigmoid, None], 'multiclass_classification', torch.float32),
                            ([10, 48, 6], [torch.relu, None], 'multiclass_classification', torch.float64),
                            ([4, 3, 3], [torch.sigmoid, None], 'multiclass_classification', torch.float64),
                            ([5, 24, 12, 1], [torch.sigmoid, torch.tanh, torch.sigmoid], 'binary_classification', torch.float32)):
    N = 120
    x = rng.standard_normal((N, dims[0]))
    y = np.eye(dims[-1])[rng.integers(0, dims[-1], N)] if lik.startswith('multi') else (rng.random((N, 1)) < 0.5).astype(float)
    data = XYDataset(torch.tensor(x, dtype=dt, device=DEV), torch.tensor(y, dtype=dt, device=DEV))
    loader = DataLoader(data, batch_size=N, shuffle=False)
    model = mlp.MLP(loss=loss_functions[lik], hparams=mlp.Hyperparameters(dims=dims, activations=acts), dtype=dt, device=DEV)
    P = model.num_params()
    model.prior = Normal(torch.zeros(P, dtype=dt, device=DEV), torch.ones(P, dtype=dt, device=DEV))
    C = 96
    th0 = 0.1 * torch.randn(C, P, dtype=dt, device=DEV)
    kern = model._plan(*next(iter(loader))).kernel
    res = []
    for S, kw in ((HMC, dict(step=0.01, num_steps=5)), (MALA, dict(step=1e-4)), (MetropolisHastings, {})):
        s = S(model, theta0=th0, dataloader=loader, seed=3, **kw)
        if S is MetropolisHastings:
            s.kernel.set_density_params(s.current['sample'], scale=torch.full((P,), 0.005, dtype=dt, device=DEV))
        if S is HMC:
            s.tuner = PerChainDATuner(torch.full((C,), 0.01, dtype=torch.float64, device=DEV), num_steps=5)
        st = ChainStats(C, P, DEV); st.attach(model._plan(*next(iter(loader))))
        s.run(num_epochs=30, num_burnin_epochs=10)
        ch = s.get_chain()
        assert ch.get_samples().shape == (20, C, P) and torch.isfinite(ch.get_target_vals()).all()
        summ = st.summary()
        res.append(f"{S.__name__} acc {ch.acceptance_rate().mean().item():.2f}")
        st.detach(model._plan(*next(iter(loader)))) if hasattr(st, 'detach') else None
    print(dims, str(dt)[6:], kern, ' | '.join(res))

# ---- the reference's own way of using the package: ONE chain, torch generator, ChainList storage, scalar HMCDATuner;
# minibatches from a shuffling DataLoader; the parallel-tempering sampler; posterior predictive; chain files
import tempfile
from eeyore_amd.chains import ChainList
from eeyore_amd.samplers import PowerPosteriorSampler
from eeyore_amd.tuners import HMCDATuner
for dims, acts, lik, dt in (([4, 16, 3], [torch.sigmoid, None], 'multiclass_classification', torch.float64),
                            ([10, 100, 10], [torch.sigmoid, None], 'multiclass_classification', torch.float32),
                            ([4, 3, 3], [torch.sigmoid, None], 'multiclass_classification', torch.float64),
                            ([4, 32, 32, 3], [torch.sigmoid, torch.sigmoid, None], 'multiclass_classification', torch.float32)):
    N = 90
    x = rng.standard_normal((N, dims[0]))
    y = np.eye(dims[-1])[rng.integers(0, dims[-1], N)]
    data = XYDataset(torch.tensor(x, dtype=dt, device=DEV), torch.tensor(y, dtype=dt, device=DEV))
    model = mlp.MLP(loss=loss_functions[lik], hparams=mlp.Hyperparameters(dims=dims, activations=acts), dtype=dt, device=DEV)
    P = model.num_params()
    model.prior = Normal(torch.zeros(P, dtype=dt, device=DEV), torch.ones(P, dtype=dt, device=DEV))
    full = DataLoader(data, batch_size=N, shuffle=True)
    kern = model._plan(*next(iter(full))).kernel
    notes = []
    # one chain, recorded torch randomness, scalar dual averaging during burn-in
    s = HMC(model, theta0=0.1 * torch.randn(P, dtype=dt, device=DEV), dataloader=full, tuner=HMCDATuner(0.08, e0=0.02),
            chain=ChainList())
    s.run(num_epochs=30, num_burnin_epochs=10)
    ch = s.get_chain()
    assert len(ch) == 20 and torch.isfinite(torch.stack(ch.vals['target_val'])).all()
    notes.append(f"1-chain HMC acc {ch.acceptance_rate():.2f} step {float(s.step):.3g}")
    with tempfile.TemporaryDirectory() as d:
        ch.to_chainfile(path=d, mode='w')
        assert sorted(os.listdir(d)) == ['accepted.csv', 'sample.csv', 'target_val.csv'], os.listdir(d)
    # minibatches: three batches per epoch from a shuffling loader, 8 chains
    mini = DataLoader(data, batch_size=30, shuffle=True)
    s = MALA(model, theta0=0.1 * torch.randn(8, P, dtype=dt, device=DEV), dataloader=mini, step=1e-4, seed=2)
    s.run(num_epochs=6, num_burnin_epochs=2)
    assert s.get_chain().get_samples().shape[0] == 12, s.get_chain().get_samples().shape
    # ... and batches of unequal size (40, 40, 10 rows) through HMC: the plan's data images follow every change
    ragged = DataLoader(data, batch_size=40, shuffle=True)
    s = HMC(model, theta0=0.1 * torch.randn(8, P, dtype=dt, device=DEV), dataloader=ragged, step=0.005, num_steps=3, seed=2)
    s.run(num_epochs=5, num_burnin_epochs=1)
    assert s.get_chain().get_samples().shape[0] == 12 and torch.isfinite(s.get_chain().get_target_vals()).all()
    notes.append("minibatch MALA / ragged HMC ok")
    # parallel tempering: 4 temperatures x 5 replicas
    pt = PowerPosteriorSampler(model, full, [['HMC', {'step': 0.01, 'num_steps': 3}] for _ in range(4)],
                               theta0=0.1 * torch.randn(5, P, dtype=dt, device=DEV), between_step=2, seed=4)
    pt.run(num_epochs=12, num_burnin_epochs=4)
    assert pt.get_chain().get_samples().shape == (8, 5, P)
    notes.append(f"PT swaps {sum(int(sw.sum()) for _, sw, _ in pt.last_swaps)}")
    # posterior predictive over the stored samples of the multi-chain run
    samples = pt.get_chain().get_samples().reshape(-1, P)
    est, dropped = model.predictive_posterior_batched(samples, data.x[:9], data.y[:9])
    assert est.shape == (9,) and torch.isfinite(est).all() and int(dropped.sum()) == 0
    notes.append(f"predictive mean {est.mean().item():.2f}")
    print(dims, str(dt)[6:], kern, ' | '.join(notes))
