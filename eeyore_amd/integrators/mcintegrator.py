"""Monte Carlo integration over stored samples behind the reference's names ``Integrator`` / ``MCIntegrator``
(eeyore/integrators/mcintegrator.py:10-63): the running mean of f(sample, x, y) with NaN integrands dropped and
counted.  ``integrate_batched`` evaluates all samples in one chain-batched call into the HIP library."""
import torch
from torch.utils.data import DataLoader


class Integrator:
    def integrate(self, *args, **kwargs):
        raise NotImplementedError


class MCIntegrator(Integrator):
    def __init__(self, f=None, samples=None):
        self.f = f
        self.samples = samples

    def integrate(self, x, y):
        """(estimate, number of dropped samples); the estimate is the mean of the non-NaN integrands, accumulated as
        a running mean in sample order."""
        kept, dropped, mean = 0, 0, 0.
        for sample in self.samples:
            value = self.f(sample, x, y)
            if torch.isnan(value):
                dropped += 1
                continue
            kept += 1
            mean = mean + (value - mean) / kept
        return mean, dropped

    def integrate_batched(self, x, y):
        """The same estimate with ``f`` called once on the [S, P] stack of samples (``f`` must accept a batch, as the
        models' ``set_params_and_lik`` does)."""
        stack = self.samples if torch.is_tensor(self.samples) else torch.stack(list(self.samples))
        values = self.f(stack, x, y)
        ok = ~torch.isnan(values)
        return values[ok].mean(), int((~ok).sum().item())

    def integrate_from_dataset(self, dataset, num_points, shuffle=True, dtype=torch.float64, device='cpu',
                               verbose=False, verbose_step=1):
        """Integrate at ``num_points`` data points drawn one at a time from ``dataset`` (items (x, y) or (x, y, index)).
        Returns (integrals, indices, numbers of dropped samples)."""
        integrals = torch.empty(num_points, dtype=dtype, device=device)
        indices = torch.full((num_points,), -1, dtype=torch.int64, device=device)
        dropped = torch.empty(num_points, dtype=torch.int64, device=device)
        done = 0
        while done < num_points:
            for item in DataLoader(dataset, batch_size=1, shuffle=shuffle):
                if done == num_points:
                    break
                x, y = item[0], item[1]
                value, n_dropped = self.integrate(x, y)
                integrals[done] = float(value)
                if len(item) > 2:
                    indices[done] = int(item[2])
                dropped[done] = n_dropped
                done += 1
                if verbose and done % verbose_step == 0:
                    print(f"Iteration {done} out of {num_points}")
        return integrals, indices, dropped
