from .base import Model, LogTargetModel, BayesianModel
from .mlp import MLP, Hyperparameters
from . import mlp
