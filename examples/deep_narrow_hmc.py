"""HMC over the weights of a DEEPER net -- MLP(16-32-32-32-3), three hidden layers, 16 inputs -- for thousands of chains at once.

`eeyore/models/mlp.py:37-43` builds any depth and width; shapes like this one (more than two hidden layers, more than 16
inputs) are taken by the fused kernel `k_mid32` (eeyore_amd/csrc/ey_mid.hip: a workgroup per chain, one wave per row tile,
the chain's weights resident in LDS): value and gradient of every chain in one launch per leapfrog step.  The script is the
reference's HMC workflow (examples/samplers) on a seeded synthetic classification set; EEYORE_EXAMPLE_CHAINS /
EEYORE_EXAMPLE_EPOCHS shrink the run.
"""
import os
import sys
import time

import numpy as np
import torch
from torch.distributions import Normal
from torch.utils.data import DataLoader

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout
from eeyore_amd.constants import loss_functions
from eeyore_amd.datasets import XYDataset
from eeyore_amd.models import mlp
from eeyore_amd.samplers import HMC

DEVICE = 'cuda:0'
DIMS, NUM_STEPS, STEP, ROWS = [16, 32, 32, 32, 3], 10, 0.01, 150


def main():
    num_chains = int(os.environ.get('EEYORE_EXAMPLE_CHAINS', 4096))
    epochs = int(os.environ.get('EEYORE_EXAMPLE_EPOCHS', 220))
    rng = np.random.default_rng(0)
    centres = rng.standard_normal((DIMS[-1], DIMS[0]))
    labels = np.arange(ROWS) % DIMS[-1]
    x = (centres[labels] + 0.7 * rng.standard_normal((ROWS, DIMS[0]))).astype(np.float32)
    y = np.eye(DIMS[-1], dtype=np.float32)[labels]
    data = XYDataset(torch.tensor(x, device=DEVICE), torch.tensor(y, device=DEVICE))
    loader = DataLoader(data, batch_size=len(data), shuffle=False)
    model = mlp.MLP(loss=loss_functions['multiclass_classification'],
                    hparams=mlp.Hyperparameters(dims=DIMS, bias=4 * [True],
                                                activations=[torch.sigmoid, torch.sigmoid, torch.sigmoid, None]),
                    dtype=torch.float32, device=DEVICE)
    P = model.num_params()
    model.prior = Normal(torch.zeros(P, device=DEVICE), torch.ones(P, device=DEVICE))
    sampler = HMC(model, theta0=0.1 * torch.randn(num_chains, P, device=DEVICE), dataloader=loader, step=STEP,
                  num_steps=NUM_STEPS, seed=1)
    t0 = time.perf_counter()
    sampler.run(num_epochs=epochs, num_burnin_epochs=epochs // 11)
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    print(f"MLP({'-'.join(map(str, DIMS))}), {P} parameters, {num_chains} chains, kernel family: {model._plan(*next(iter(loader))).kernel}")
    print(f"Time taken: {seconds:.2f} s  ->  {num_chains * NUM_STEPS * epochs / seconds:.3e} leapfrog-steps/sec x chains")
    chain = sampler.get_chain()
    print(f"Stored samples per chain: {len(chain)}")
    print(f"Mean acceptance rate: {chain.acceptance_rate().mean().item():.3f}")


if __name__ == '__main__':
    main()
