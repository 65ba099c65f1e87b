import numpy as np
import torch

from pathlib import Path

from .chain import Chain
from eeyore_amd.constants import torch_to_np_types


class ChainFile(Chain):
    """Monte Carlo chain appended to one CSV per key, '%.18e' rows (eeyore/chains/chain_file.py:9-81)."""

    def __init__(self, keys=['sample', 'target_val', 'accepted'], path=Path.cwd(), mode='a'):
        self.path = Path(path)
        self.mode = mode
        if not self.path.exists():
            self.path.mkdir(parents=True, exist_ok=True)
        self.reset(keys=keys)

    def reset(self, keys=['sample', 'target_val', 'accepted']):
        self.vals = {key: open(self.path.joinpath(key+'.csv'), self.mode) for key in keys}

    def close(self):
        for key in self.vals.keys():
            self.vals[key].close()

    def update(self, state, reset=True, close=True,
               fmt={'sample': '%.18e', 'target_val': '%.18e', 'grad_val': '%.18e', 'accepted': '%d'}):
        if reset:
            self.reset(keys=self.vals.keys())
        for key in self.vals.keys():
            if isinstance(state[key], torch.Tensor):
                np.savetxt(self.vals[key], state[key].detach().cpu().numpy().ravel()[np.newaxis], fmt=fmt[key],
                           delimiter=',')
            elif isinstance(state[key], np.ndarray):
                np.savetxt(self.vals[key], state[key].ravel()[np.newaxis], fmt=fmt[key], delimiter=',')
            else:
                self.vals[key].write(str(state[key])+'\n')
        if close:
            self.close()

    def line_to_val_element(self, line, key, dtype=torch.float64, device='cpu'):
        if key == 'target_val':
            return torch.tensor(torch_to_np_types[dtype](line.strip())).to(device=device)
        elif key in ('sample', 'grad_val'):
            return torch.tensor(list(map(torch_to_np_types[dtype], line.split(',')))).to(device=device)
        elif key == 'accepted':
            return int(line.strip())

    def to_chainlist(self, keys=None, dtype=torch.float64, device='cpu'):
        from .chain_list import ChainList

        keys = set(keys or self.vals.keys()) & set(['sample', 'target_val', 'grad_val', 'accepted'])
        vals = {}
        for key in keys:
            with open(self.path.joinpath(key+'.csv'), mode='r') as file:
                vals[key] = [self.line_to_val_element(line, key, dtype=dtype, device=device) for line in file.readlines()]
        return ChainList(vals=vals)
