# MALA sampling of MLP weights using Iris data on an MI355X, chain stored in a list.
#
# Counterpart of the reference's examples/samplers/mlp/iris/mala_gpu_chainlist.py with `eeyore` -> `eeyore_amd`
# (plotting and kanga post-processing left out).  EEYORE_EXAMPLE_EPOCHS shortens the run for tests.

import os
import torch

from datetime import timedelta
from timeit import default_timer as timer
from torch.distributions import Normal
from torch.utils.data import DataLoader

from eeyore_amd.constants import loss_functions
from eeyore_amd.datasets import XYDataset
from eeyore_amd.models import mlp
from eeyore_amd.samplers import MALA

device = 'cuda:0'
num_epochs = int(os.environ.get('EEYORE_EXAMPLE_EPOCHS', 11000))
num_burnin_epochs = num_epochs // 11

# %% Load Iris data

iris = XYDataset.from_eeyore('iris', yndmin=1, dtype=torch.float32, device=device, yonehot=True)
dataloader = DataLoader(iris, batch_size=len(iris), shuffle=True)

# %% Setup MLP model

hparams = mlp.Hyperparameters(dims=[4, 3, 3], activations=[torch.sigmoid, None])
model = mlp.MLP(
    loss=loss_functions['multiclass_classification'],
    hparams=hparams,
    dtype=torch.float32,
    device=device
)
model.prior = Normal(
    torch.zeros(model.num_params(), dtype=model.dtype, device=device),
    (3 * torch.ones(model.num_params(), dtype=model.dtype, device=device)).sqrt()
)

# %% Setup MALA sampler

sampler = MALA(
    model,
    theta0=model.prior.sample(),
    dataloader=dataloader,
    step=0.003
)

# %% Run MALA sampler

start_time = timer()

sampler.run(num_epochs=num_epochs, num_burnin_epochs=num_burnin_epochs, verbose=True, verbose_step=max(1, num_epochs // 11))

end_time = timer()
print("Time taken: {}".format(timedelta(seconds=end_time-start_time)))

# %% Summaries from the ChainList

chain = sampler.get_chain()
print('Number of stored samples: {}'.format(len(chain)))
print('Acceptance rate: {}'.format(chain.acceptance_rate()))
print('Monte Carlo mean: {}'.format(chain.mean()))
