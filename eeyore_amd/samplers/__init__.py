from .sampler import Sampler
from .serial_sampler import SerialSampler
from .single_chain_serial_sampler import SingleChainSerialSampler
from .hmc import HMC
from .mala import MALA
from .metropolis_hastings import MetropolisHastings
from .power_posterior_sampler import PowerPosteriorSampler
