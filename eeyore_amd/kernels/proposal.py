"""Proposal densities behind the reference's names (``Kernel``, ``NormalizedKernel``, ``NormalKernel``;
eeyore/kernels/{kernel,normalized_kernel,normal_kernel}.py).

Inside the samplers the proposal draw and its log-density are part of the fused HIP step (``ey_mala_step`` /
``ey_mh_step``); a ``NormalKernel`` object is how a script hands the proposal scale to ``MetropolisHastings`` and how
it can inspect or evaluate the proposal density of the current state."""
import torch
from torch.distributions import Normal


class Kernel:
    """k(x1, x2): a kernel evaluated at a pair of points."""

    def k(self, x1, x2):
        raise NotImplementedError


class NormalizedKernel(Kernel):
    """A kernel that is a probability density in its first argument: wraps a torch distribution as ``density``;
    ``log_prob`` sums the elementwise log-densities (normalized_kernel.py:14-15)."""

    density = None

    def log_prob(self, state):
        return self.density.log_prob(state).sum()

    def sample(self):
        return self.density.sample()


class NormalKernel(NormalizedKernel):
    """Independent normals N(loc_i, scale_i)."""

    def __init__(self, loc, scale):
        self.set_density(loc, scale)

    def set_density(self, loc, scale):
        self.density = Normal(loc, scale)

    def set_density_params(self, loc, scale=None):
        """Re-centre (and optionally re-scale) the existing density in place."""
        self.density.loc = loc
        if scale is not None:
            self.density.scale = scale

    def k(self, x1, x2, scale=None):
        """Density of x1 under the kernel centred at x2."""
        self.set_density_params(x2, scale=scale)
        return torch.exp(self.log_prob(x1))
