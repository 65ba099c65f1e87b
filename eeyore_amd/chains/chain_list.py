import torch

from pathlib import Path

import eeyore_amd.stats as st

from .chain import Chain


class ChainList(Chain):
    """Monte Carlo chain stored as python lists, one entry per saved iteration (eeyore/chains/chain_list.py:12-141).
    Summary statistics delegate to eeyore_amd.stats (chain_list.py:69-102)."""

    def __init__(self, keys=['sample', 'target_val', 'accepted'], vals=None):
        self.reset(keys=keys, vals=vals)

    def reset(self, keys=['sample', 'target_val', 'accepted'], vals=None):
        if vals is None:
            self.vals = {key: [] for key in keys}
        else:
            self.vals = vals

    def __repr__(self):
        return f"Markov chain containing {len(self)} samples."

    def __len__(self):
        return self.num_samples()

    def num_params(self):
        return len(self.get_sample(0))

    def num_samples(self):
        return len(self.vals['sample'])

    def get_param(self, idx):
        return torch.stack([sample[idx] for sample in self.vals['sample']])

    def get_sample(self, idx):
        return self.vals['sample'][idx]

    def get_samples(self):
        return torch.stack(self.vals['sample'])

    def get_target_vals(self):
        return torch.stack(self.vals['target_val'])

    def get_grad_val(self, idx):
        return self.vals['grad_val'][idx]

    def get_grad_vals(self):
        return torch.stack(self.vals['grad_val'])

    def state(self, idx=-1):
        current = {}
        for key, val in self.vals.items():
            try:
                current[key] = val[idx]
            except IndexError:
                print(f'WARNING: chain does not have values for {key}.')
        return current

    def update(self, state):
        for key in self.vals.keys():
            self.vals[key].append(state[key])

    def mean(self):
        return self.get_samples().mean(0)

    def running_mean(self, idx):
        return st.running_mean(self.get_param(idx))

    def running_means(self):
        return st.running_mean(self.get_samples(), dim=0)

    def mc_se(self, mc_cov_mat=None, method='inse', adjust=False):
        if mc_cov_mat is None:
            return st.mc_se(self.get_samples(), method=method, adjust=adjust, rowvar=False)
        return st.mc_se_from_cov(mc_cov_mat)

    def mc_cov(self, method='inse', adjust=False):
        return st.mc_cov(self.get_samples(), method=method, adjust=adjust, rowvar=False)

    def mc_cor(self, mc_cov_mat=None, method='inse', adjust=False):
        if mc_cov_mat is None:
            return st.mc_cor(self.get_samples(), method=method, adjust=adjust, rowvar=False)
        return st.cor_from_cov(mc_cov_mat)

    def multi_ess(self, mc_cov_mat=None, method='inse', adjust=False):
        return st.multi_ess(self.get_samples(), mc_cov_mat=mc_cov_mat, method=method, adjust=adjust)

    def block_acceptance_rate(self):
        return torch.stack(self.vals['accepted']).sum(axis=0) / self.num_samples()

    def acceptance_rate(self):
        """Proportion of accepted samples: sum(accepted) / num_samples (chain_list.py:94-96)."""
        return sum(self.vals['accepted']) / self.num_samples()

    def save(self, path):
        torch.save(self.vals, path)

    def load(self, path):
        self.vals = torch.load(path)

    def to_chainfile(self, keys=None, path=Path.cwd(), mode='a',
                     fmt={'sample': '%.18e', 'target_val': '%.18e', 'grad_val': '%.18e', 'accepted': '%d'}):
        from .chain_file import ChainFile

        chainfile = ChainFile(keys=keys or self.vals.keys(), path=path, mode=mode)
        for i in range(len(self)):
            chainfile.update(self.state(i), reset=False, close=False, fmt=fmt)
        chainfile.close()
