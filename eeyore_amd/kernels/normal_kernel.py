from torch.distributions import Normal

from .normalized_kernel import NormalizedKernel


class NormalKernel(NormalizedKernel):
    """Normal proposal density (eeyore/kernels/normal_kernel.py:5-23).  MALA and MetropolisHastings read its
    ``density.loc`` / ``density.scale``; the proposal draw and its log-density are evaluated inside the fused
    HIP step (ey_mala_step / ey_mh_step)."""

    def __init__(self, loc, scale):
        self.set_density(loc, scale)

    def set_density(self, loc, scale):
        self.density = Normal(loc, scale)

    def set_density_params(self, loc, scale=None):
        self.density.loc = loc
        if scale is not None:
            self.density.scale = scale

    def k(self, x1, x2, scale=None):
        self.set_density_params(x2, scale=scale)
        return self.log_prob(x1).exp()
