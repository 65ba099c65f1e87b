"""Thousands of HMC chains over the weights of MLP(4-32-32-3) on Iris, advanced together on one MI355X.

The reference runs one chain per sampler; here ``theta0`` of shape [C, P] makes each ``draw`` one fused HIP launch
(momentum draw, L leapfrog steps, accept) for all C chains.  EEYORE_EXAMPLE_CHAINS / EEYORE_EXAMPLE_EPOCHS shrink the run.
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
from eeyore_amd.distributed import ChainStats
from eeyore_amd.models import mlp
from eeyore_amd.samplers import HMC

DEVICE = 'cuda:0'
NUM_STEPS, STEP = 20, 0.024


def main():
    num_chains = int(os.environ.get('EEYORE_EXAMPLE_CHAINS', 4096))
    epochs = int(os.environ.get('EEYORE_EXAMPLE_EPOCHS', 1100))
    iris = XYDataset.from_eeyore('iris', yndmin=1, yonehot=True, dtype=torch.float32, device=DEVICE)
    loader = DataLoader(iris, batch_size=len(iris), shuffle=False)
    model = mlp.MLP(loss=loss_functions['multiclass_classification'],
                    hparams=mlp.Hyperparameters(dims=[4, 32, 32, 3], bias=3 * [True],
                                                activations=[torch.sigmoid, torch.sigmoid, None]),
                    dtype=torch.float32, device=DEVICE)
    P = model.num_params()
    model.prior = Normal(torch.zeros(P, device=DEVICE), torch.full((P,), 3.0, device=DEVICE).sqrt())

    sampler = HMC(model, theta0=0.1 * torch.randn(num_chains, P, device=DEVICE), dataloader=loader, step=STEP,
                  num_steps=NUM_STEPS, seed=1)
    t0 = time.perf_counter()
    sampler.run(num_epochs=epochs, num_burnin_epochs=epochs // 11)
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    print(f"Time taken: {seconds:.2f} s  ->  {num_chains * NUM_STEPS * epochs / seconds:.3e} leapfrog-steps/sec x chains")

    chain = sampler.get_chain()  # ChainBuffer: [iters, C, P] on the device
    print(f"Stored samples per chain: {len(chain)}")
    print(f"Mean acceptance rate: {chain.acceptance_rate().mean().item():.3f}")
    stats = ChainStats(num_chains, P, DEVICE)
    for i in range(len(chain)):
        stats.update(chain.get_samples()[i].contiguous(), chain.get_accepted()[i].contiguous())
    print(f"max R-hat over parameters: {stats.summary()['rhat'].max().item():.3f}")
    print(f"chain 0 as a ChainList: {chain.get_chain(0)}")


if __name__ == '__main__':
    main()
