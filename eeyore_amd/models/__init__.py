from .base import Model, LogTargetModel, BayesianModel
from .mlp import MLP, Hyperparameters
from . import mlp
from .logistic_regression import LogisticRegression
from . import logistic_regression
