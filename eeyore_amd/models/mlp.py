"""``mlp.Hyperparameters`` and ``mlp.MLP`` with the reference's constructors (eeyore/models/mlp.py:9-50).

The layers are ``nn.Linear`` modules so that ``parameters()`` order, ``state_dict`` and ``forward`` are what scripts
written for the reference expect; the sampler hot path never runs them -- it goes through ``_plan(x, y)``, the C-ABI plan
that carries this model's dims / bias flags / activation codes / likelihood code, its prior and the data.
"""
import torch
import torch.nn as nn
from torch.distributions import Normal

from eeyore_amd.plan import Plan

from .base import BayesianModel

# activation -> kernel code (include/eeyore_amd.h: enum ey_act)
_ACTIVATIONS = (
    (0, (None,), ()),
    (1, (torch.sigmoid, torch.nn.functional.sigmoid), (nn.Sigmoid,)),
    (2, (torch.tanh, torch.nn.functional.tanh), (nn.Tanh,)),
    (3, (torch.relu, torch.nn.functional.relu), (nn.ReLU,)),
)


def activation_code(fn):
    for code, callables, module_types in _ACTIVATIONS:
        if any(fn is c for c in callables) or (module_types and isinstance(fn, module_types)):
            return code
    raise ValueError(f"activation {fn!r} has no HIP kernel (supported: None, torch.sigmoid, torch.tanh, torch.relu)")


class Hyperparameters:
    """Layer widths, per-layer bias flags and activations; at least one hidden layer and one activation per layer,
    otherwise ``ValueError`` (mlp.py:15-19)."""

    def __init__(self, dims=[1, 2, 1], bias=None, activations=None):
        n_layers = len(dims) - 1
        self.dims = dims
        self.bias = n_layers * [True] if bias is None else bias
        self.activations = n_layers * [torch.sigmoid] if activations is None else activations
        if len(self.dims) < 3 or len(self.activations) != n_layers or len(self.bias) != n_layers:
            raise ValueError


class MLP(BayesianModel):
    def __init__(self, loss, temperature=None, prior=None, hparams=Hyperparameters(), savefile=None,
                 dtype=torch.float64, device='cpu'):
        super().__init__(loss, temperature=temperature, dtype=dtype, device=device)
        self.hp = hparams
        self.fc_layers = self.set_fc_layers()
        self._hip_plan = None
        self.prior = prior or self.default_prior()
        if savefile:
            self.load_state_dict(torch.load(savefile), strict=False)

    # `model.prior = Normal(...)` after construction is the idiom of the reference's examples: a setter lets the plan
    # notice the change and re-upload the prior
    @property
    def prior(self):
        return self._prior

    @prior.setter
    def prior(self, value):
        object.__setattr__(self, "_prior", value)
        object.__setattr__(self, "_prior_uploaded", False)

    def default_prior(self):
        """N(0, 1) on every parameter (mlp.py:31-35)."""
        shape = (self.num_params(),)
        return Normal(torch.zeros(shape, dtype=self.dtype, device=self.device),
                      torch.ones(shape, dtype=self.dtype, device=self.device))

    def set_fc_layers(self):
        widths = self.hp.dims
        return nn.ModuleList(
            nn.Linear(widths[k], widths[k + 1], bias=self.hp.bias[k]).to(dtype=self.dtype, device=self.device)
            for k in range(len(widths) - 1))

    def forward(self, x):
        for layer, act in zip(self.fc_layers, self.hp.activations):
            x = layer(x)
            x = x if act is None else act(x)
        return x

    def num_hidden_layers(self):
        return len(self.hp.dims) - 2

    def _plan(self, x, y):
        """This model's C-ABI plan with the current prior and, when given, the (x, y) batch attached."""
        plan = self._hip_plan
        if plan is None:
            code = getattr(self.loss, "code", None)
            if code is None:
                raise ValueError("loss must be one of eeyore_amd.constants.loss_functions (the kernels implement "
                                 "BCE-sum on probabilities and CE-sum on logits)")
            acts = [activation_code(a) for a in self.hp.activations]
            plan = Plan(self.hp.dims, self.hp.bias, acts, code, self.dtype, self.device)
            object.__setattr__(self, "_hip_plan", plan)
        if not self._prior_uploaded:
            if not isinstance(self._prior, Normal):
                raise ValueError("only an elementwise torch.distributions.Normal prior has a HIP kernel")
            plan.set_prior(self._prior.loc, self._prior.scale)
            object.__setattr__(self, "_prior_uploaded", True)
        if x is not None:
            plan.set_data(x, y)
        return plan
