"""After the run: convergence diagnostics and the posterior predictive for a multi-chain HMC run on Iris.

Per-parameter R-hat over all chains (running moments kept by the step kernels), the effective sample size of every
(chain, parameter) series with the reference's initial-sequence estimator in one device pass, and the posterior
predictive probability of held-out points integrated over the pooled samples.
EEYORE_EXAMPLE_CHAINS / EEYORE_EXAMPLE_EPOCHS shrink the run.
"""
import os
import sys

import torch
from torch.distributions import Normal
from torch.utils.data import DataLoader

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout
from eeyore_amd.constants import loss_functions
from eeyore_amd.datasets import XYDataset
from eeyore_amd.distributed import ChainStats, reduce_ess
from eeyore_amd.models import mlp
from eeyore_amd.samplers import HMC

DEVICE = 'cuda:0'


def main():
    num_chains = int(os.environ.get('EEYORE_EXAMPLE_CHAINS', 512))
    epochs = int(os.environ.get('EEYORE_EXAMPLE_EPOCHS', 660))
    iris = XYDataset.from_eeyore('iris', yndmin=1, yonehot=True, dtype=torch.float32, device=DEVICE)
    held_out = torch.arange(0, len(iris), 10, device=DEVICE)           # every tenth flower
    keep = torch.ones(len(iris), dtype=torch.bool, device=DEVICE)
    keep[held_out] = False
    train = XYDataset(iris.x[keep], iris.y[keep])
    loader = DataLoader(train, batch_size=len(train), shuffle=False)
    model = mlp.MLP(loss=loss_functions['multiclass_classification'],
                    hparams=mlp.Hyperparameters(dims=[4, 32, 32, 3], bias=3 * [True],
                                                activations=[torch.sigmoid, torch.sigmoid, None]),
                    dtype=torch.float32, device=DEVICE)
    P = model.num_params()
    model.prior = Normal(torch.zeros(P, device=DEVICE), torch.full((P,), 3.0, device=DEVICE).sqrt())

    sampler = HMC(model, theta0=0.1 * torch.randn(num_chains, P, device=DEVICE), dataloader=loader, step=0.024,
                  num_steps=20, seed=1)
    burnin = epochs // 11
    # the step kernels keep the running chain moments from the first stored iteration on
    sampler.run(num_epochs=burnin, num_burnin_epochs=burnin)
    stats = ChainStats(num_chains, P, DEVICE)
    stats.attach(model._plan(train.x, train.y))
    sampler.run(num_epochs=epochs, num_burnin_epochs=burnin)
    chain = sampler.get_chain()
    print(f"Stored samples per chain: {len(chain)}; mean acceptance rate: {chain.acceptance_rate().mean().item():.3f}")
    summary = stats.summary()
    print(f"R-hat over {summary['num_chains']} chains: max {summary['rhat'].max().item():.3f}, "
          f"median {summary['rhat'].median().item():.3f}")

    ess = reduce_ess(chain.ess())                                        # [C, P] -> per-parameter figures
    print(f"ESS per chain and parameter: min {ess['min'].min().item():.1f}, mean {ess['mean'].mean().item():.1f} "
          f"of {len(chain)} iterations ({ess['not_enough']} series too short to estimate)")

    pooled = chain.get_samples()[-20:].reshape(-1, P)                    # the last 20 iterations of every chain
    probs, dropped = model.predictive_posterior_batched(pooled, iris.x[held_out], iris.y[held_out])
    print(f"Posterior predictive probability of the true class of {len(held_out)} held-out flowers "
          f"({pooled.shape[0]} samples each, {int(dropped.sum())} dropped): "
          f"mean {probs.mean().item():.3f}, min {probs.min().item():.3f}")
    print(f"Mean acceptance rate: {chain.acceptance_rate().mean().item():.3f}")


if __name__ == '__main__':
    main()
