import torch

from .log_target_model import LogTargetModel


class BayesianModel(LogTargetModel):
    """eeyore/models/bayesian_model.py:6-67: log_lik = -loss(forward(x), y), log_prior = sum prior.log_prob(theta),
    both multiplied by ``temperature`` when it is set.  Sub-classes provide ``_plan()``."""

    def __init__(self, loss, temperature=None, dtype=torch.float64, device='cpu'):
        super().__init__(temperature=temperature, dtype=dtype, device=device)
        self.loss = loss

    def default_prior(self):
        raise NotImplementedError

    def summary(self, hashsummary=False):
        print(self)
        print("-" * 80)
        print(f"Number of model parameters: {self.num_params()}")
        print("-" * 80)
        print(f"Prior: {self.prior}")
        print("-" * 80)
        if hashsummary:
            print('Hash Summary:')
            for idx, hashvalue in enumerate(self.hashsummary()):
                print(f"{idx}: {hashvalue}")

    # ---- hot path: every method below is one call into the HIP library
    def _batched(self, theta):
        th = theta.detach()
        single = th.dim() == 1
        th = (th[None] if single else th).to(device=self.device, dtype=self.dtype).contiguous()
        return th, single

    def log_lik(self, x, y):
        """Log-likelihood of the model's current parameters (bayesian_model.py:30-35)."""
        return self.set_params_and_log_lik(self.get_params(), x, y, _set=False)

    def set_params_and_log_lik(self, theta, x, y, _set=True):
        if _set:
            self.set_params(theta if theta.dim() == 1 else theta[0])
        plan = self._plan(x, y)
        th, single = self._batched(theta)
        lik, _ = plan.log_target(th, temp=self.temperature)
        return lik[0] if single else lik

    def set_params_and_lik(self, theta, x, y):
        return torch.exp(self.set_params_and_log_lik(theta, x, y))

    def log_prior(self, theta=None):
        """bayesian_model.py:46-50 (the reference evaluates it at the model's current parameters)."""
        plan = self._plan(None, None)
        th, single = self._batched(self.get_params() if theta is None else theta)
        _, prior = plan.log_target(th, temp=self.temperature, prior_only=True)
        return prior[0] if single else prior

    def log_target(self, theta, x, y):
        """bayesian_model.py:52-56.  theta [P] -> 0-d tensor; theta [C, P] -> [C]."""
        self.set_params(theta if theta.dim() == 1 else theta[0])
        plan = self._plan(x, y)
        th, single = self._batched(theta)
        lik, prior = plan.log_target(th, temp=self.temperature)
        t = lik + prior
        return t[0] if single else t

    def upto_grad_log_target(self, theta, x, y):
        """log_target_model.py:20-23.  theta [P] -> (0-d, [P]); theta [C, P] -> ([C], [C, P])."""
        self.set_params(theta if theta.dim() == 1 else theta[0])
        plan = self._plan(x, y)
        th, single = self._batched(theta)
        t, g = plan.log_target_grad(th, temp=self.temperature)
        return (t[0], g[0]) if single else (t, g)

    def predictive_posterior(self, theta, x, y):
        from eeyore_amd.integrators import MCIntegrator
        integrator = MCIntegrator(f=lambda s, x, y: self.set_params_and_lik(s.clone().detach(), x, y), samples=theta)
        return integrator.integrate(x, y)

    def predictive_posterior_from_dataset(self, theta, dataset, num_points, shuffle=True, verbose=False, verbose_step=1):
        from eeyore_amd.integrators import MCIntegrator
        integrator = MCIntegrator(f=lambda s, x, y: self.set_params_and_lik(s.clone().detach(), x, y), samples=theta)
        return integrator.integrate_from_dataset(dataset, num_points, shuffle=shuffle, verbose=verbose,
                                                 verbose_step=verbose_step)
