import torch

from .storage import Chain, ChainList, ChainLists


class _Offload:
    """A device-to-host copy in flight (ChainBuffer.offload_async)."""

    def __init__(self, host, event):
        self._host, self._event = host, event

    def done(self):
        return self._event.query()

    def wait(self):
        self._event.synchronize()
        return self._host


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

    def rewind(self):
        """Forget the stored iterations but keep the device buffers (a second run of the same length records into the same
        memory: no allocation inside it)."""
        self.n = 0

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

    # ---- off the device without stalling the sampler
    def offload_async(self, start=0, stop=None, stream=None):
        """Start copying iterations [start, stop) of every stored key to pinned host memory on a side stream and return
        a handle; the step kernels keep running on the sampler's stream meanwhile (the copy only waits for what was
        recorded before this call).  ``handle.wait()`` returns {key: host tensor}; ``handle.done()`` polls.
        A chain of 1000 x 4096 x 1315 floats is 21.5 GB: offloading block by block while the next block is sampled keeps
        the device buffer small (``handle.wait()`` then ``drop_front(stop)``)."""
        stop = self.n if stop is None else min(stop, self.n)
        dev = self.bufs['sample'].device
        side = stream or torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))   # what has been recorded so far
        host = {}
        with torch.cuda.stream(side):
            for k in self.keys:
                src = self.bufs[k][start:stop]
                dst = torch.empty(src.shape, dtype=src.dtype, device='cpu', pin_memory=True)
                dst.copy_(src, non_blocking=True)
                src.record_stream(side)
                host[k] = dst
            event = torch.cuda.Event()
            event.record(side)
        return _Offload(host, event)

    def drop_front(self, count):
        """Forget the first ``count`` stored iterations (after they have been offloaded): the rest moves to the front."""
        count = min(count, self.n)
        for k in self.keys:
            self.bufs[k][:self.n - count] = self.bufs[k][count:self.n].clone()
        self.n -= count

    def to_chainfiles(self, path, mode='w', chains=None):
        """``runNN``-style directories of the reference's CSV files, one per chain (eeyore/chains/chain_file.py:21-45):
        the whole buffer goes to the host in one asynchronous copy, the files are written from there."""
        from pathlib import Path
        host = self.offload_async().wait()
        cs = range(self.num_chains()) if chains is None else chains
        width = len(str(self.num_chains()))
        for c in cs:
            vals = {}
            for k in self.keys:
                col = host[k][:, c]
                vals[k] = [int(a) for a in col.tolist()] if k == 'accepted' else list(col.unbind(0))
            ChainList(keys=self.keys, vals=vals).to_chainfile(path=Path(path) / f'run{str(c + 1).zfill(width)}', mode=mode)

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
