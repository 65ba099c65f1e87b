"""Chain storage behind the reference's names (``Chain``, ``ChainList``, ``ChainLists``, ``ChainFile``).

The reference keeps one python list per state key and grows it by one entry per saved iteration
(eeyore/chains/chain_list.py:64-67); scripts read ``chain.vals[key]`` directly, so that attribute is kept.  Everything
else here is organised around two small helpers: ``_stacked`` (turn a key's history into one tensor) and the CSV row
codec used by ``ChainFile`` (one file per key, one row per iteration, '%.18e' for reals and '%d' for the accept flag,
eeyore/chains/chain_file.py:28-45).  Summary statistics delegate to ``eeyore_amd.stats``.
"""
from pathlib import Path

import numpy as np
import torch

import eeyore_amd.stats as st
from eeyore_amd.constants import torch_to_np_types

DEFAULT_KEYS = ('sample', 'target_val', 'accepted')
CSV_FORMATS = {'sample': '%.18e', 'target_val': '%.18e', 'grad_val': '%.18e', 'accepted': '%d'}
_VECTOR_KEYS = ('sample', 'grad_val')


class Chain:
    """What a sampler needs from a chain: ``update(state)`` and ``reset``; ``detach_and_update`` snapshots tensors first
    so later in-place changes of the sampler state cannot alter history (eeyore/chains/chain.py:12-13)."""

    def reset(self):
        raise NotImplementedError

    def update(self, state):
        raise NotImplementedError

    def detach_and_update(self, state):
        snapshot = {}
        for key, value in state.items():
            snapshot[key] = value.detach().clone() if torch.is_tensor(value) else value
        self.update(snapshot)


class ChainList(Chain):
    """In-memory chain: ``vals[key]`` is the list of that key's values, one per saved iteration."""

    def __init__(self, keys=DEFAULT_KEYS, vals=None):
        self.reset(keys=keys, vals=vals)

    def reset(self, keys=DEFAULT_KEYS, vals=None):
        self.vals = {k: [] for k in keys} if vals is None else vals

    def update(self, state):
        for key, history in self.vals.items():
            history.append(state[key])

    # -- sizes
    def num_samples(self):
        return len(self.vals['sample'])

    __len__ = num_samples

    def num_params(self):
        return len(self.vals['sample'][0])

    def __repr__(self):
        return f"Markov chain containing {self.num_samples()} samples."

    # -- element and bulk access
    def _stacked(self, key):
        return torch.stack(self.vals[key])

    def get_sample(self, idx):
        return self.vals['sample'][idx]

    def get_grad_val(self, idx):
        return self.vals['grad_val'][idx]

    def get_samples(self):
        return self._stacked('sample')

    def get_target_vals(self):
        return self._stacked('target_val')

    def get_grad_vals(self):
        return self._stacked('grad_val')

    def get_param(self, idx):
        return self._stacked('sample')[:, idx]

    def state(self, idx=-1):
        picked = {}
        for key, history in self.vals.items():
            if -len(history) <= idx < len(history):
                picked[key] = history[idx]
            else:
                print(f'WARNING: chain does not have values for {key}.')
        return picked

    # -- summaries (chain_list.py:69-102)
    def mean(self):
        return self.get_samples().mean(0)

    def running_mean(self, idx):
        return st.running_mean(self.get_param(idx))

    def running_means(self):
        return st.running_mean(self.get_samples(), dim=0)

    def mc_cov(self, method='inse', adjust=False):
        return st.mc_cov(self.get_samples(), method=method, adjust=adjust, rowvar=False)

    def mc_se(self, mc_cov_mat=None, method='inse', adjust=False):
        cov_mat = self.mc_cov(method=method, adjust=adjust) if mc_cov_mat is None else mc_cov_mat
        return st.mc_se_from_cov(cov_mat)

    def mc_cor(self, mc_cov_mat=None, method='inse', adjust=False):
        cov_mat = self.mc_cov(method=method, adjust=adjust) if mc_cov_mat is None else mc_cov_mat
        return st.cor_from_cov(cov_mat)

    def multi_ess(self, mc_cov_mat=None, method='inse', adjust=False):
        return st.multi_ess(self.get_samples(), mc_cov_mat=mc_cov_mat, method=method, adjust=adjust)

    def acceptance_rate(self):
        """sum(accepted) / num_samples (chain_list.py:94-96)."""
        return sum(self.vals['accepted']) / self.num_samples()

    def block_acceptance_rate(self):
        return self._stacked('accepted').sum(axis=0) / self.num_samples()

    # -- persistence
    def save(self, path):
        torch.save(self.vals, path)

    def load(self, path):
        self.vals = torch.load(path)

    def to_chainfile(self, keys=None, path=Path.cwd(), mode='a', fmt=CSV_FORMATS):
        out = ChainFile(keys=list(keys or self.vals.keys()), path=path, mode=mode)
        for i in range(self.num_samples()):
            out.update(self.state(i), reset=False, close=False, fmt=fmt)
        out.close()


def _csv_row(value, fmt):
    """One CSV line for a state entry: tensors/arrays are flattened, python scalars written as they print."""
    if torch.is_tensor(value):
        value = value.detach().cpu().numpy()
    if isinstance(value, np.ndarray):
        return ','.join(fmt % v for v in value.reshape(-1)) + '\n'
    return f'{value}\n'


def _parse_row(line, key, np_type, device):
    if key == 'accepted':
        return int(line)
    if key in _VECTOR_KEYS:
        return torch.tensor(np.array(line.split(','), dtype=np_type), device=device)
    return torch.tensor(np_type(line), device=device)


class ChainFile(Chain):
    """Chain appended to ``<path>/<key>.csv``, one row per saved iteration (eeyore/chains/chain_file.py:9-81)."""

    def __init__(self, keys=DEFAULT_KEYS, path=Path.cwd(), mode='a'):
        self.path = Path(path)
        self.mode = mode
        self.path.mkdir(parents=True, exist_ok=True)
        self.reset(keys=keys)

    def _file(self, key):
        return self.path / f'{key}.csv'

    def reset(self, keys=DEFAULT_KEYS):
        self.vals = {key: open(self._file(key), self.mode) for key in list(keys)}

    def close(self):
        for handle in self.vals.values():
            handle.close()

    def update(self, state, reset=True, close=True, fmt=CSV_FORMATS):
        if reset:
            self.reset(keys=self.vals.keys())
        for key, handle in self.vals.items():
            handle.write(_csv_row(state[key], fmt.get(key, '%s')))
        if close:
            self.close()

    def to_chainlist(self, keys=None, dtype=torch.float64, device='cpu'):
        wanted = [k for k in (keys or self.vals.keys()) if k in CSV_FORMATS]
        np_type = torch_to_np_types[dtype]
        vals = {}
        for key in wanted:
            with open(self._file(key)) as handle:
                vals[key] = [_parse_row(line.strip(), key, np_type, device) for line in handle if line.strip()]
        return ChainList(vals=vals)


class ChainLists:
    """Several chains side by side: ``vals[key][chain]`` is that chain's history list (chain_lists.py:7-155)."""

    def __init__(self, keys=DEFAULT_KEYS, vals=None):
        self.reset(keys=keys, vals=vals)

    def reset(self, keys=DEFAULT_KEYS, vals=None):
        self.vals = {k: [] for k in keys} if vals is None else vals

    @classmethod
    def from_chain_list(cls, chain_lists, keys=DEFAULT_KEYS):
        shared = set(keys)
        for chain in chain_lists:
            shared &= set(chain.vals.keys())
        return cls(keys=shared, vals={k: [chain.vals[k] for chain in chain_lists] for k in shared})

    @classmethod
    def from_file(cls, paths, keys=DEFAULT_KEYS, mode='a', dtype=torch.float64, device='cpu'):
        loaded = [ChainFile(keys=keys, path=p, mode=mode).to_chainlist(dtype=dtype, device=device) for p in paths]
        return cls.from_chain_list(loaded, keys=keys)

    def num_chains(self):
        return len(self.vals['sample'])

    __len__ = num_chains

    def num_samples(self):
        return len(self.vals['sample'][0])

    def num_params(self):
        return len(self.vals['sample'][0][0])

    def __repr__(self):
        return f"{self.num_chains()} Markov chains, each containing {self.num_samples()} samples."

    def get_chain(self, idx, key='sample'):
        return torch.stack(self.vals[key][idx])

    def _all(self, key):
        return torch.stack([self.get_chain(i, key=key) for i in range(self.num_chains())])

    def get_samples(self):
        return self._all('sample')

    def get_target_vals(self):
        return self._all('target_val')

    def get_grad_vals(self):
        return self._all('grad_val')

    # -- per-chain statistics and their summaries
    def _per_chain(self, fn):
        return [fn(i, self.get_chain(i, key='sample')) for i in range(self.num_chains())]

    def mean(self):
        return self.get_samples().mean(1)

    def mean_summary(self, g=lambda x: torch.mean(x, dim=0)):
        return g(self.mean())

    def mc_cov(self, method='inse', adjust=False):
        return torch.stack(self._per_chain(lambda i, x: st.mc_cov(x, method=method, adjust=adjust, rowvar=False)))

    def mc_cov_summary(self, g=lambda m: torch.mean(m, dim=0), method='inse', adjust=False):
        return g(self.mc_cov(method=method, adjust=adjust))

    def mc_se(self, mc_cov_mat=None, method='inse', adjust=False):
        covs = self.mc_cov(method=method, adjust=adjust) if mc_cov_mat is None else mc_cov_mat
        return torch.stack([st.mc_se_from_cov(covs[i]) for i in range(self.num_chains())])

    def mc_se_summary(self, g=lambda x: torch.mean(x, dim=0), mc_cov_mat=None, method='inse', adjust=False):
        return g(self.mc_se(mc_cov_mat=mc_cov_mat, method=method, adjust=adjust))

    def mc_cor(self, mc_cov_mat=None, method='inse', adjust=False):
        covs = self.mc_cov(method=method, adjust=adjust) if mc_cov_mat is None else mc_cov_mat
        return torch.stack([st.cor_from_cov(covs[i]) for i in range(self.num_chains())])

    def mc_cor_summary(self, g=lambda m: torch.mean(m, dim=0), mc_cov_mat=None, method='inse', adjust=False):
        return g(self.mc_cor(mc_cov_mat=mc_cov_mat, method=method, adjust=adjust))

    def acceptance(self):
        n = self.num_samples()
        return [sum(history) / n for history in self.vals['accepted']]

    def acceptance_summary(self, g=lambda x: sum(x) / len(x)):
        return g(self.acceptance())

    def multi_ess(self, mc_cov_mat=None, method='inse', adjust=False):
        return self._per_chain(lambda i, x: st.multi_ess(
            x, mc_cov_mat=None if mc_cov_mat is None else mc_cov_mat[i], method=method, adjust=adjust))

    def multi_ess_summary(self, g=lambda x: sum(x) / len(x), mc_cov_mat=None, method='inse', adjust=False):
        return g(self.multi_ess(mc_cov_mat=mc_cov_mat, method=method, adjust=adjust))

    def multi_rhat(self, mc_cov_mat=None, method='inse', adjust=False):
        return st.multi_rhat(self.get_samples(), mc_cov_mat=mc_cov_mat, method=method, adjust=adjust)

    def summary(self, keys=('multi_ess', 'multi_rhat'), g_mean_summary=lambda x: torch.mean(x, dim=0),
                g_mc_se_summary=lambda x: torch.mean(x, dim=0), g_acceptance_summary=lambda x: sum(x) / len(x),
                g_multi_ess_summary=lambda x: sum(x) / len(x), mc_cov_mat=None, method='inse', adjust=False):
        """Selected summaries in one pass; the per-chain MC covariances are computed once and shared
        (chain_lists.py:125-155)."""
        if mc_cov_mat is None and {'mc_se', 'multi_ess', 'multi_rhat'} & set(keys):
            mc_cov_mat = self.mc_cov(method=method, adjust=adjust)
        shared = dict(mc_cov_mat=mc_cov_mat, method=method, adjust=adjust)
        table = {
            'mean': lambda: self.mean_summary(g=g_mean_summary),
            'mc_se': lambda: self.mc_se_summary(g=g_mc_se_summary, **shared),
            'acceptance': lambda: self.acceptance_summary(g=g_acceptance_summary),
            'multi_ess': lambda: self.multi_ess_summary(g=g_multi_ess_summary, **shared),
            'multi_rhat': lambda: self.multi_rhat(**shared)[0],
        }
        return {key: table[key]() for key in keys if key in table}
