from .core import DataCounter, EmptyXYDataset, XYDataset, batches, data_paths
from . import synthetic
