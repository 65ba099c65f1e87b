import torch

from .integrator import Integrator


class MCIntegrator(Integrator):
    """Monte Carlo integration of f(sample, x, y) over stored samples with NaN integrands dropped and counted
    (eeyore/integrators/mcintegrator.py:10-30).  ``samples`` may be a list of [P] tensors or a [S, P] tensor; when
    ``f`` accepts a batch the whole integral is one call into the HIP library (``integrate_batched``)."""

    def __init__(self, f=None, samples=None):
        super().__init__()
        self.f = f
        self.samples = samples

    def integrate(self, x, y):
        integral = 0.
        num_kept_samples = 1
        num_dropped_samples = 0
        for sample in self.samples:
            integrand = self.f(sample, x, y)
            if torch.isnan(integrand):
                num_dropped_samples = num_dropped_samples + 1
            else:
                integral = ((num_kept_samples - 1) * integral + integrand) / num_kept_samples
                num_kept_samples = num_kept_samples + 1
        return integral, num_dropped_samples

    def integrate_batched(self, x, y):
        """Same estimate with all samples evaluated in one chain-batched call: mean of the non-NaN integrands."""
        samples = self.samples if isinstance(self.samples, torch.Tensor) else torch.stack(list(self.samples))
        vals = self.f(samples, x, y)
        keep = ~torch.isnan(vals)
        return vals[keep].mean(), int((~keep).sum().item())
