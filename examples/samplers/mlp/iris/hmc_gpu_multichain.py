# HMC sampling of MLP(4-32-32-3) weights on Iris with thousands of chains advanced together on one MI355X.
#
# The reference has no multi-chain HMC; this is the same script shape as its single-chain examples with a [C, P]
# starting point.  Each `draw` is one fused HIP launch (momentum draw, L leapfrog steps, accept) for all chains.

import os
import torch

from datetime import timedelta
from timeit import default_timer as timer
from torch.distributions import Normal
from torch.utils.data import DataLoader

from eeyore_amd.constants import loss_functions
from eeyore_amd.datasets import XYDataset
from eeyore_amd.distributed import ChainStats
from eeyore_amd.models import mlp
from eeyore_amd.samplers import HMC

device = 'cuda:0'
num_chains = int(os.environ.get('EEYORE_EXAMPLE_CHAINS', 4096))
num_epochs = int(os.environ.get('EEYORE_EXAMPLE_EPOCHS', 1100))
num_burnin_epochs = num_epochs // 11

iris = XYDataset.from_eeyore('iris', yndmin=1, dtype=torch.float32, device=device, yonehot=True)
dataloader = DataLoader(iris, batch_size=len(iris), shuffle=False)

hparams = mlp.Hyperparameters(dims=[4, 32, 32, 3], bias=3*[True], activations=[torch.sigmoid, torch.sigmoid, None])
model = mlp.MLP(loss=loss_functions['multiclass_classification'], hparams=hparams, dtype=torch.float32, device=device)
P = model.num_params()
model.prior = Normal(torch.zeros(P, device=device), (3 * torch.ones(P, device=device)).sqrt())

theta0 = 0.1 * torch.randn(num_chains, P, device=device)
sampler = HMC(model, theta0=theta0, dataloader=dataloader, step=0.024, num_steps=20, seed=1)

start_time = timer()
sampler.run(num_epochs=num_epochs, num_burnin_epochs=num_burnin_epochs)
torch.cuda.synchronize()
runtime = timer() - start_time
print("Time taken: {}".format(timedelta(seconds=runtime)))
print("leapfrog-steps/sec x chains: {:.3e}".format(num_chains * 20 * num_epochs / runtime))

chain = sampler.get_chain()                      # ChainBuffer [iters, C, P] on the device
print('Stored samples per chain: {}'.format(len(chain)))
print('Mean acceptance rate: {:.3f}'.format(chain.acceptance_rate().mean().item()))

stats = ChainStats(num_chains, P, device)
for i in range(len(chain)):
    stats.update(chain.get_samples()[i].contiguous(), chain.get_accepted()[i].contiguous())
summary = stats.summary()
print('max R-hat over parameters: {:.3f}'.format(summary['rhat'].max().item()))
print('chain 0 as a reference-style ChainList: {}'.format(sampler.get_chain().get_chain(0)))
