from .data_counter import DataCounter
from .data_info import data_paths
from .empty_dataset import EmptyXYDataset
from .xydataset import XYDataset
from . import synthetic
