import torch

from .xydataset import XYDataset


class EmptyXYDataset(XYDataset):
    """eeyore/datasets/empty_dataset.py:5-7."""

    def __init__(self, dtype=torch.float64, device='cpu'):
        super().__init__(torch.tensor([[]], dtype=dtype, device=device), torch.tensor([[]], dtype=dtype, device=device))

    def __repr__(self):
        return 'Empty XYDataset'
