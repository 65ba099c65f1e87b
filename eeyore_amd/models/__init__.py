from .model import Model
from .log_target_model import LogTargetModel
from .bayesian_model import BayesianModel
from .mlp import MLP, Hyperparameters
from . import mlp
