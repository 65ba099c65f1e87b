#!/usr/bin/env python3
"""The sampler surface end to end on every kernel family and both dtypes: HMC with a per-chain dual-averaging tuner, MALA and
random-walk MH, chain statistics attached, burn-in then stored iterations.  usage: python tools/soak_samplers.py"""
import sys, numpy as np, torch
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
