"""``logistic_regression.Hyperparameters`` and ``LogisticRegression`` with the reference's constructors
(eeyore/models/logistic_regression.py:8-37): one ``nn.Linear`` and an activation (sigmoid by default), i.e. the one-layer
case of the MLP plan -- log-likelihood, log-target, gradient, the samplers and the posterior predictive all run through the
same C-ABI kernels."""
import torch
import torch.nn as nn
from torch.distributions import Normal

from eeyore_amd.plan import Plan

from .base import BayesianModel
from .mlp import activation_code


class Hyperparameters:
    def __init__(self, input_size=1, output_size=1, bias=True, activation=torch.sigmoid):
        self.input_size = input_size
        self.output_size = output_size
        self.bias = bias
        self.activation = activation


class LogisticRegression(BayesianModel):
    def __init__(self, loss, temperature=None, prior=None, hparams=Hyperparameters(), savefile=None,
                 dtype=torch.float64, device='cpu'):
        super().__init__(loss, temperature=temperature, dtype=dtype, device=device)
        self.hp = hparams
        self.linear = nn.Linear(self.hp.input_size, self.hp.output_size, bias=self.hp.bias).to(
            dtype=self.dtype, device=self.device)
        self._hip_plan = None
        self.prior = prior or self.default_prior()
        if savefile:
            self.load_state_dict(torch.load(savefile), strict=False)

    @property
    def prior(self):
        return self._prior

    @prior.setter
    def prior(self, value):  # `model.prior = Normal(...)` after construction re-uploads the prior (as mlp.MLP)
        object.__setattr__(self, "_prior", value)
        object.__setattr__(self, "_prior_uploaded", False)

    def default_prior(self):
        """N(0, 1) on every parameter (logistic_regression.py:27-31)."""
        shape = (self.num_params(),)
        return Normal(torch.zeros(shape, dtype=self.dtype, device=self.device),
                      torch.ones(shape, dtype=self.dtype, device=self.device))

    def forward(self, x):
        x = self.linear(x)
        return x if self.hp.activation is None else self.hp.activation(x)

    def _plan(self, x, y):
        """This model's C-ABI plan (a one-layer MLP) with the current prior and, when given, the (x, y) batch."""
        plan = self._hip_plan
        if plan is None:
            code = getattr(self.loss, "code", None)
            if code is None:
                raise ValueError("loss must be one of eeyore_amd.constants.loss_functions (the kernels implement "
                                 "BCE-sum on probabilities and CE-sum on logits)")
            plan = Plan([self.hp.input_size, self.hp.output_size], [bool(self.hp.bias)],
                        [activation_code(self.hp.activation)], code, self.dtype, self.device)
            object.__setattr__(self, "_hip_plan", plan)
        if not self._prior_uploaded:
            if not isinstance(self._prior, Normal):
                raise ValueError("only an elementwise torch.distributions.Normal prior has a HIP kernel")
            plan.set_prior(self._prior.loc, self._prior.scale)
            object.__setattr__(self, "_prior_uploaded", True)
        if x is not None:
            plan.set_data(x, y)
        return plan
