from .core import DataCounter, EmptyXYDataset, XYDataset, data_paths
from . import synthetic
