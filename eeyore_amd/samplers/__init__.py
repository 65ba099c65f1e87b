from .base import Sampler, SerialSampler, SingleChainSerialSampler
from .hmc import HMC
from .mala import MALA
from .metropolis_hastings import MetropolisHastings
from .power_posterior_sampler import PowerPosteriorSampler
