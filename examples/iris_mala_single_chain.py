"""One MALA chain over the weights of MLP(4-3-3) on Iris, on an MI355X.

What a script written for the reference's single-chain API looks like after switching the imports to eeyore_amd:
the model, prior, DataLoader, sampler constructor and ``run`` call are the reference's; only the device differs.
Set EEYORE_EXAMPLE_EPOCHS to shorten the run.
"""
import os
import sys
import time

import torch
from torch.distributions import Normal
from torch.utils.data import DataLoader

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout
from eeyore_amd.constants import loss_functions
from eeyore_amd.datasets import XYDataset
from eeyore_amd.models import mlp
from eeyore_amd.samplers import MALA

DEVICE = 'cuda:0'
DTYPE = torch.float32


def build_model():
    net = mlp.MLP(loss=loss_functions['multiclass_classification'],
                  hparams=mlp.Hyperparameters(dims=[4, 3, 3], activations=[torch.sigmoid, None]),
                  dtype=DTYPE, device=DEVICE)
    n = net.num_params()
    net.prior = Normal(torch.zeros(n, dtype=DTYPE, device=DEVICE), torch.full((n,), 3.0, dtype=DTYPE, device=DEVICE).sqrt())
    return net


def main():
    epochs = int(os.environ.get('EEYORE_EXAMPLE_EPOCHS', 11000))
    burnin = epochs // 11
    iris = XYDataset.from_eeyore('iris', yndmin=1, yonehot=True, dtype=DTYPE, device=DEVICE)
    loader = DataLoader(iris, batch_size=len(iris), shuffle=True)
    model = build_model()
    sampler = MALA(model, theta0=model.prior.sample(), dataloader=loader, step=0.003)

    t0 = time.perf_counter()
    sampler.run(num_epochs=epochs, num_burnin_epochs=burnin, verbose=True, verbose_step=max(1, epochs // 11))
    print(f"Time taken: {time.perf_counter() - t0:.2f} s")

    chain = sampler.get_chain()
    print(f"Stored samples: {len(chain)}")
    print(f"Acceptance rate: {chain.acceptance_rate():.3f}")
    print(f"Monte Carlo mean: {chain.mean()}")
    if len(chain) >= 200:
        # in f64: the 27-dimensional determinants of multi_ess underflow in f32 (the reference's multi_ess would too)
        from eeyore_amd.stats import multi_ess
        print(f"Multivariate ESS: {multi_ess(chain.get_samples().double()):.1f}")


if __name__ == '__main__':
    main()
