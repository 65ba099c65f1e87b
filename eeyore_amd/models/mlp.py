import torch
import torch.nn as nn
from torch.distributions import Normal

from eeyore_amd.plan import Plan

from .bayesian_model import BayesianModel

_ACT_CODES = {None: 0, torch.sigmoid: 1, torch.tanh: 2, torch.relu: 3}


def activation_code(a):
    if a in _ACT_CODES:
        return _ACT_CODES[a]
    if isinstance(a, nn.Sigmoid) or a is torch.nn.functional.sigmoid:
        return 1
    if isinstance(a, nn.Tanh) or a is torch.nn.functional.tanh:
        return 2
    if isinstance(a, nn.ReLU) or a is torch.nn.functional.relu:
        return 3
    raise ValueError(f"activation {a!r} has no HIP kernel (supported: None, torch.sigmoid, torch.tanh, torch.relu)")


class Hyperparameters:
    """eeyore/models/mlp.py:9-19."""

    def __init__(self, dims=[1, 2, 1], bias=None, activations=None):
        self.dims = dims
        self.bias = bias if bias is not None else (len(dims) - 1) * [True]
        self.activations = activations if activations is not None else (len(dims) - 1) * [torch.sigmoid]

        if len(self.dims) < 3:
            raise ValueError

        if (len(self.dims) != len(self.activations)+1):
            raise ValueError

        if (len(self.bias) != len(self.activations)):
            raise ValueError


class MLP(BayesianModel):
    """eeyore/models/mlp.py:21-50.  Same constructor; ``device`` must name an MI355X for the hot path."""

    def __init__(self, loss, temperature=None, prior=None, hparams=Hyperparameters(), savefile=None,
                 dtype=torch.float64, device='cpu'):
        super().__init__(loss, temperature=temperature, dtype=dtype, device=device)
        self.hp = hparams
        self.fc_layers = self.set_fc_layers()
        self._hip_plan = None
        self._prior = None
        self._prior_dirty = True
        self.prior = prior or self.default_prior()
        if savefile:
            self.load_state_dict(torch.load(savefile), strict=False)

    # `model.prior = Normal(...)` after construction is the idiom of the reference examples
    @property
    def prior(self):
        return self._prior

    @prior.setter
    def prior(self, value):
        object.__setattr__(self, "_prior", value)
        object.__setattr__(self, "_prior_dirty", True)

    def default_prior(self):
        return Normal(
            torch.zeros(self.num_params(), dtype=self.dtype, device=self.device),
            torch.ones(self.num_params(), dtype=self.dtype, device=self.device)
        )

    def set_fc_layers(self):
        fc = []
        for i in range(len(self.hp.dims)-1):
            fc.append(nn.Linear(
                self.hp.dims[i], self.hp.dims[i+1], bias=self.hp.bias[i]
            ).to(dtype=self.dtype, device=self.device))
        return nn.ModuleList(fc)

    def forward(self, x):
        for fc, activation in zip(self.fc_layers, self.hp.activations):
            x = fc(x)
            if activation is not None:
                x = activation(x)
        return x

    def num_hidden_layers(self):
        return len(self.hp.dims)-2

    # ---- HIP plan plumbing
    def _plan(self, x, y):
        """The C-ABI plan of this model with (x, y) and the current prior attached."""
        if self._hip_plan is None:
            code = getattr(self.loss, "code", None)
            if code is None:
                raise ValueError("loss must be one of eeyore_amd.constants.loss_functions (the kernels implement "
                                 "BCE-sum on probabilities and CE-sum on logits)")
            acts = [activation_code(a) for a in self.hp.activations]
            object.__setattr__(self, "_hip_plan", Plan(self.hp.dims, self.hp.bias, acts, code, self.dtype, self.device))
        plan = self._hip_plan
        if self._prior_dirty:
            pr = self._prior
            if not isinstance(pr, Normal):
                raise ValueError("only an elementwise torch.distributions.Normal prior has a HIP kernel")
            plan.set_prior(pr.loc, pr.scale)
            object.__setattr__(self, "_prior_dirty", False)
        if x is not None:
            plan.set_data(x, y)
        return plan
