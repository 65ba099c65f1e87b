import torch

from .storage import Chain, ChainList, ChainLists


class ChainBuffer(Chain):
    """Chain storage for C chains advanced together: device buffers ``sample [iters, C, P]``,
    ``target_val [iters, C]``, ``accepted [iters, C]`` grown geometrically, written by one ``copy_`` per saved
    iteration (no host sync).  ``get_chain(c)`` / ``to_chainlists()`` give the reference's per-chain views
    (eeyore/chains/chain_list.py, chain_lists.py)."""

    def __init__(self, keys=['sample', 'target_val', 'accepted'], capacity=0):
        self.keys = list(keys)
        self.capacity = capacity
        self.reset()

    def reset(self, keys=None):
        if keys is not None:
            self.keys = list(keys)
        self.n = 0
        self.bufs = {}

    @property
    def vals(self):
        return {k: None for k in self.keys}

    def __len__(self):
        return self.n

    def num_samples(self):
        return self.n

    def reserve(self, num_iters, state):
        for k in self.keys:
            v = state[k]
            cur = self.bufs.get(k)
            if cur is None or cur.shape[0] < num_iters:
                new = torch.empty((num_iters,) + tuple(v.shape), dtype=v.dtype, device=v.device)
                if cur is not None and self.n:
                    new[:self.n] = cur[:self.n]
                self.bufs[k] = new

    def update(self, state):
        need = self.n + 1
        any_buf = next(iter(self.bufs.values()), None)
        if any_buf is None or any_buf.shape[0] < need:
            self.reserve(max(need, self.capacity, 2 * (0 if any_buf is None else any_buf.shape[0])), state)
        for k in self.keys:
            self.bufs[k][self.n].copy_(state[k])
        self.n += 1

    def block(self, k, state):
        """Writable views of the next k iterations of every stored key (after reserving room), for a kernel that
        records k iterations in one launch; ``commit(k)`` makes them part of the chain."""
        any_buf = next(iter(self.bufs.values()), None)
        need = self.n + k
        if any_buf is None or any_buf.shape[0] < need:
            self.reserve(max(need, self.capacity, 2 * (0 if any_buf is None else any_buf.shape[0])), state)
        return {key: self.bufs[key][self.n:self.n + k] for key in self.keys}

    def commit(self, k):
        self.n += k

    def detach_and_update(self, state):
        self.update(state)  # update() copies into the buffer, so no clone is needed

    # ---- accessors
    def num_chains(self):
        return self.bufs['sample'].shape[1]

    def num_params(self):
        return self.bufs['sample'].shape[2]

    def get_samples(self):
        """[iters, C, P]"""
        return self.bufs['sample'][:self.n]

    def get_target_vals(self):
        return self.bufs['target_val'][:self.n]

    def get_accepted(self):
        return self.bufs['accepted'][:self.n]

    def mean(self):
        """Per-chain Monte Carlo means [C, P]."""
        return self.get_samples().mean(0)

    def ess(self):
        """Effective sample size of every (chain, parameter) series, [C, P] (one device pass, stats.batched.ess)."""
        from eeyore_amd.stats import batched
        return batched.ess(self.get_samples())

    def mc_se(self):
        """Monte Carlo standard error (the reference's mc_se, p = 1) of every (chain, parameter) series, [C, P]."""
        from eeyore_amd.stats import batched
        return batched.mc_se(self.get_samples())

    def acceptance_rate(self):
        """Per-chain acceptance [C] = sum(accepted) / num_samples (chain_list.py:94-96)."""
        return self.get_accepted().to(torch.float64).mean(0)

    def get_chain(self, c):
        """Chain c as a reference-style ChainList (host-visible python lists of tensors)."""
        vals = {}
        for k in self.keys:
            col = self.bufs[k][:self.n, c]
            if k == 'accepted':
                vals[k] = [int(a) for a in col.cpu().tolist()]
            else:
                vals[k] = list(col.unbind(0))
        return ChainList(keys=self.keys, vals=vals)

    def to_chainlists(self):
        return ChainLists.from_chain_list([self.get_chain(c) for c in range(self.num_chains())], keys=self.keys)
