"""Parallel-tempered HMC over the weights of MLP(784-128-10) on MNIST-shaped data (BASELINE config 5), through the
sampler surface.

P = 101 770 parameters per chain do not fit a CU's LDS: ``plan.kernel`` is ``bgemm`` -- every layer of every chain is a
tile job of a chain-batched GEMM, the leapfrog update rides in the epilogues of the gradient kernels.  One temperature per
GPU (``torchrun --nproc-per-node K examples/mnist_shaped_tempering.py``; one process = one temperature = plain HMC),
replicas exchange temperature LABELS between neighbouring ranks every few iterations (``distributed.TemperingExchange``),
never their 100 k-float states.  On ONE GPU, EEYORE_EXAMPLE_LADDER=K runs the same algorithm with all K temperatures in this
process (``distributed.LocalTemperingLadder``: K x chains as one chain batch with a per-chain temperature vector).
EEYORE_EXAMPLE_CHAINS / EEYORE_EXAMPLE_EPOCHS / EEYORE_EXAMPLE_ROWS size the run.
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
from eeyore_amd.distributed import LocalTemperingLadder, TemperingExchange, init_from_env
from eeyore_amd.models import mlp
from eeyore_amd.samplers import HMC

NUM_STEPS, STEP, EXCHANGE_EVERY = 20, 0.001, 5


def main():
    rank, world, local = init_from_env()
    device = torch.device('cuda', local % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    num_chains = int(os.environ.get('EEYORE_EXAMPLE_CHAINS', 256))
    epochs = int(os.environ.get('EEYORE_EXAMPLE_EPOCHS', 20))
    rows = int(os.environ.get('EEYORE_EXAMPLE_ROWS', 1024))

    rng = np.random.default_rng(0)  # MNIST-shaped: 784 pixels, ~19 % of them non-zero, ten balanced classes
    x = (rng.random((rows, 784)) * (rng.random((rows, 784)) < 0.19)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[np.arange(rows) % 10]
    data = XYDataset(torch.tensor(x, device=device), torch.tensor(y, device=device))
    loader = DataLoader(data, batch_size=rows, shuffle=False)

    model = mlp.MLP(loss=loss_functions['multiclass_classification'],
                    hparams=mlp.Hyperparameters(dims=[784, 128, 10], activations=[torch.sigmoid, None]),
                    dtype=torch.float32, device=device)
    P = model.num_params()
    model.prior = Normal(torch.zeros(P, device=device), torch.ones(P, device=device))

    local_ladder = int(os.environ.get('EEYORE_EXAMPLE_LADDER', 1)) if world == 1 else 1
    temps = max(world, local_ladder)
    ladder = [(i / temps) ** 4 for i in range(1, temps + 1)]   # the reference's default power-posterior ladder
    if local_ladder > 1:   # every temperature in this process: local_ladder x num_chains chains in one batch
        exchange = LocalTemperingLadder(ladder, num_chains, device, seed=11)
        num_chains *= local_ladder
    else:
        exchange = TemperingExchange(ladder, num_chains, rank, world, device, seed=11)
    sampler = HMC(model, theta0=0.05 * torch.randn(num_chains, P, device=device), dataloader=loader, step=STEP,
                  num_steps=NUM_STEPS, seed=1 + rank, temperature=exchange.temperature_vector(torch.float32))
    if rank == 0:
        print(f"kernel family: {model._plan(*next(iter(loader))).kernel}; {temps} temperature(s) x "
              f"{num_chains // local_ladder} chains, P = {P}")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    swaps, accepted, stored = 0, 0.0, 0
    for start in range(0, epochs, EXCHANGE_EVERY):   # run() starts a fresh chain record, as the reference's does
        sampler.run(num_epochs=min(EXCHANGE_EVERY, epochs - start), num_burnin_epochs=0)
        accepted += sampler.get_chain().get_accepted().float().mean(1).sum().item()
        stored += len(sampler.get_chain())
        if temps > 1:
            swaps += int(exchange.exchange(sampler.current['target_val'] / sampler.temperature))  # untempered log-target
            sampler.set_temperature(exchange.temperature_vector(torch.float32))   # a relabelled replica keeps its state
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    if rank == 0:
        print(f"Time taken: {seconds:.2f} s  ->  {world * num_chains * NUM_STEPS * epochs / seconds:.3e} "
              f"leapfrog-steps/sec x chains")
        print(f"Iterations per chain: {stored}; mean acceptance rate: {accepted / stored:.3f}; "
              f"label exchanges accepted: {swaps}")


if __name__ == '__main__':
    main()
