import itertools

import numpy as np
import torch

from pathlib import Path

from .hmc import HMC
from .mala import MALA
from .metropolis_hastings import MetropolisHastings
from .base import SerialSampler
from eeyore_amd.chains import ChainBuffer, ChainFile
from eeyore_amd.datasets import DataCounter


class PowerPosteriorSampler(SerialSampler):
    """Power-posterior (parallel-tempering) sampler with the reference's ladder, partner choice and swap rule
    (eeyore/samplers/power_posterior_sampler.py:15-182), chain-batched.

    ``samplers`` is the reference's list ``[[name, kwargs], ...]``, one entry per temperature; ``name`` is 'MALA' or
    'MetropolisHastings' as in the reference (:71-82), or 'HMC' (an extension the reference does not offer).  All
    temperatures are advanced by ONE fused step: the K x R chains (K temperatures, R independent replicas of the whole
    ladder; ``theta0`` of shape [P] gives R = 1, [R, P] gives R ladders) form one chain batch whose per-chain
    temperature vector goes to the HIP kernel (temperature multiplies log-likelihood and log-prior,
    eeyore/models/bayesian_model.py:33-34,48-49).  State row ``k * R + r`` is temperature k of replica r.

    Between-chain moves follow the reference exactly: for i = 0..K-1 in order, a partner j is drawn from
    Categorical(prob ~ exp(-b|i-j|)) (:107-122) and the states of i and j are exchanged iff
    ``log(u) < log q(i|j) - log q(j|i) + (t_i - t_j)(ell(theta_j) - ell(theta_i))`` (:135-141,160; ``ell`` the
    untempered log-target, decided by ``ey_pt_swap_decide``).  An exchange needs no re-evaluation: the tempered
    target and gradient of a moved state are rescaled by t_new / t_old (the reference re-evaluates, :143-151)."""

    def __init__(self, model, dataloader, samplers, theta0=None, data0=None, counter=None, temperature=None,
                 between_step=10, b=0.5, storage='list', keys=['sample', 'target_val'], path=Path.cwd(), mode='a',
                 check_input=False, rng=None, seed=0):
        super().__init__(counter or DataCounter.from_dataloader(dataloader))
        self.between_step = between_step
        self.b = b
        self.num_chains = len(samplers)
        self.dataloader = dataloader
        self.model = model
        self.sampler_names = [samplers[i][0] for i in range(self.num_chains)]
        if len(set(self.sampler_names)) != 1:
            raise ValueError("all temperatures must use the same within-chain sampler to be advanced by one fused step")
        if storage not in ('list', 'file'):
            raise ValueError("storage must be 'list' or 'file'")
        self.storage = storage
        self.keys = list(keys)
        self.dtype, self.device = model.dtype, model.device
        K = self.num_chains
        th = theta0.detach().to(device=self.device, dtype=self.dtype)
        th = th[None] if th.dim() == 1 else th
        self.num_replicas = R = th.shape[0]
        self.set_temperature(temperature)
        tvec = torch.tensor(self.temperature, dtype=self.dtype, device=self.device).repeat_interleave(R)
        theta_all = th.repeat(K, 1).contiguous()  # row k*R + r
        name = self.sampler_names[0]
        kw = [samplers[i][1] for i in range(K)]
        common = dict(theta0=theta_all, dataloader=dataloader, data0=data0 or next(iter(dataloader)),
                      counter=self.counter, chain=ChainBuffer(keys=[]), temperature=tvec, rng=rng, seed=seed)

        def per_chain(key, default):
            vals = [float(k.get(key, default)) for k in kw]
            if len(set(vals)) == 1:
                return vals[0]
            return torch.tensor(vals, dtype=self.dtype, device=self.device).repeat_interleave(R)

        if name == 'MALA':
            self.sampler = MALA(model, step=per_chain('step', 0.1), **common)
        elif name == 'HMC':
            ns = {int(k.get('num_steps', 10)) for k in kw}
            if len(ns) != 1:
                raise ValueError("num_steps must be the same at every temperature")
            self.sampler = HMC(model, step=per_chain('step', 0.1), num_steps=ns.pop(), **common)
        elif name == 'MetropolisHastings':
            self.sampler = MetropolisHastings(model, **common)
            if any('kernel' in k for k in kw):
                self.sampler.kernel = kw[0]['kernel']
        else:
            raise ValueError(f"unknown within-chain sampler {name!r}")
        self.chains = [self.init_chain(i, storage, self.keys, Path(path), mode) for i in range(K)]
        self._tvec = tvec
        self._log_q = torch.tensor(np.log(self._partner_matrix()), dtype=self.dtype, device=self.device)
        self._probs = [torch.tensor(self.eval_categorical_probs(i), dtype=torch.float64) for i in range(K)]

    def init_chain(self, i, storage, keys, path, mode):
        """The chain of temperature i (power_posterior_sampler.py:57-66): in memory, or appended to
        ``<path>/chain<i+1>/<key>.csv`` iteration by iteration.  In memory it is a device buffer [iters, R, ...] for the R
        replicas of the ladder; on file a single ladder (R = 1) writes the reference's files, R > 1 one directory per
        replica, ``chain<i+1>/replica<r+1>``."""
        if storage == 'list':
            return ChainBuffer(keys=keys)
        folder = path / f"chain{i + 1:0{len(str(self.num_chains))}}"
        if self.num_replicas == 1:
            return ChainFile(keys=keys, path=folder, mode=mode)
        width = len(str(self.num_replicas))
        handles = [ChainFile(keys=keys, path=folder / f"replica{r + 1:0{width}}", mode=mode) for r in range(self.num_replicas)]
        # K x R x len(keys) descriptors would stay open until garbage collection (8 temperatures x 1024 replicas x 3 keys
        # against a limit of 1024): ChainFile.update reopens its files anyway, so they are closed here
        for handle in handles:
            handle.close()
        return handles

    # ---- ladder and partner distribution (power_posterior_sampler.py:84-125)
    def default_indicator(self):
        return self.num_chains - 1

    def set_temperature(self, temperature):
        if (temperature is not None) and (self.num_chains != len(temperature)):
            raise ValueError
        if temperature is None:
            self.temperature = [(i/self.num_chains)**4 for i in range(1, self.num_chains+1)]
        else:
            self.temperature = list(temperature)

    def from_seq_to_events(self, k, i):
        return k if (k < i) else (k+1)

    def from_events_to_seq(self, j, i):
        return j if (j < i) else (j-1)

    def eval_categorical_prob(self, j, i):
        eb = np.exp(-self.b)
        numerator = eb**np.absolute(j-i)
        denominator = eb*(2-eb**i-eb**(self.num_chains-1-i))/(1-eb)
        return numerator/denominator

    def eval_categorical_probs(self, i):
        return np.array([self.eval_categorical_prob(j, i)
                         for j in itertools.chain(range(i), range(i+1, self.num_chains))])

    def _partner_matrix(self):
        """Q[i, j] = probability that chain i proposes partner j (Categorical normalises its weights, :43)."""
        K = self.num_chains
        Q = np.ones((K, K))  # diagonal unused (log 1 = 0)
        for i in range(K):
            p = self.eval_categorical_probs(i)
            p = p / p.sum()
            for k, j in enumerate(itertools.chain(range(i), range(i+1, K))):
                Q[i, j] = p[k]
        return Q

    def categorical_log_prob(self, j, i):
        return self._log_q[i, j]

    # ---- reference surface
    def get_model(self, idx=None):
        return self.model

    def get_chain(self, idx=None):
        return self.chains[self.default_indicator() if idx is None else idx]

    def within_chain_moves(self, x, y):
        self.sampler.draw(x, y, savestate=False)

    def _ell(self):
        """Untempered log-targets ell = T_k / t_k of every chain, [K, R]."""
        return (self.sampler._target / self._tvec).view(self.num_chains, self.num_replicas)

    def between_chain_move_log_rate(self, i, j_idx, ell=None):
        """log-rate of exchanging chain i with partners j_idx [R] (:135-141); also returns the decision inputs."""
        ell = self._ell() if ell is None else ell
        R = self.num_replicas
        ar = torch.arange(R, device=self.device)
        t = torch.tensor(self.temperature, dtype=self.dtype, device=self.device)
        ell_i, ell_j = ell[i], ell[j_idx, ar]
        t_i, t_j = t[i].expand(R), t[j_idx]
        dlogq = self._log_q[j_idx, i] - self._log_q[i, j_idx]
        return ell_i.contiguous(), ell_j.contiguous(), t_i.contiguous(), t_j.contiguous(), dlogq.contiguous()

    def _sample_partners(self, i):
        k = torch.multinomial(self._probs[i], self.num_replicas, replacement=True)
        j = torch.where(k < i, k, k + 1)
        return j.to(self.device)

    def _rand(self, n):
        return torch.rand(n, dtype=self.dtype, device=self.device)

    def between_chain_moves(self, x, y):
        s = self.sampler
        K, R = self.num_chains, self.num_replicas
        plan = self.model._plan(x, y)
        ar = torch.arange(R, device=self.device)
        has_grad = hasattr(s, '_grad')
        self.last_swaps, self.last_swap_inputs = [], []
        for i in range(K):
            j = self._sample_partners(i)
            ell_i, ell_j, t_i, t_j, dlogq = self.between_chain_move_log_rate(i, j)
            swap, log_rate = plan.pt_swap_decide(ell_i, ell_j, t_i, t_j, self._rand(R), dlogq=dlogq)
            self.last_swaps.append((j, swap, log_rate))
            self.last_swap_inputs.append((ell_i, ell_j, t_i, t_j, dlogq))
            m = swap.bool()
            if not bool(m.any()):
                continue
            ri = (i * R + ar)[m]
            rj = (j * R + ar)[m]
            ti, tj = t_i[m], t_j[m]
            th_i, th_j = s._theta[ri].clone(), s._theta[rj].clone()
            s._theta[ri], s._theta[rj] = th_j, th_i
            tg_i, tg_j = s._target[ri].clone(), s._target[rj].clone()
            s._target[ri], s._target[rj] = tg_j * (ti / tj), tg_i * (tj / ti)
            if has_grad:
                g_i, g_j = s._grad[ri].clone(), s._grad[rj].clone()
                s._grad[ri], s._grad[rj] = g_j * (ti / tj)[:, None], g_i * (tj / ti)[:, None]

    def save_state(self, i):
        s, R = self.sampler, self.num_replicas
        sl = slice(i * R, (i + 1) * R)
        state = {'sample': s._theta[sl], 'target_val': s._target[sl]}
        if 'grad_val' in self.keys:
            state['grad_val'] = s._grad[sl]
        if 'accepted' in self.keys:
            state['accepted'] = s.current['accepted'][sl]
        chain = self.chains[i]
        if self.storage == 'list':
            chain.update(state)
            return
        # on file: one row per saved iteration and replica, as ChainFile.update writes a single chain's state; the [R, ...]
        # state crosses to the host once per key, not once per replica
        host = {k: v.detach().cpu() for k, v in state.items()}
        for r, handle in enumerate([chain] if R == 1 else chain):
            handle.update({k: (int(v[r]) if k == 'accepted' else v[r]) for k, v in host.items()})

    def draw(self, x, y, savestate=False):
        """power_posterior_sampler.py:174-182."""
        self.within_chain_moves(x, y)

        if ((self.counter.idx % self.between_step) == 0):
            self.between_chain_moves(x, y)

        if savestate:
            for i in range(self.num_chains):
                self.save_state(i)

    # ---- the multi-chain surface of the reference (eeyore/samplers/multi_chain_serial_sampler.py:10-46); `chain_idx` is
    #      a temperature (default: the last one, the target itself, power_posterior_sampler.py:84-85)
    def _memory_chain(self, idx):
        chain = self.get_chain(idx)
        if not isinstance(chain, ChainBuffer):
            raise RuntimeError("this sampler stores its chains on file (storage='file'): read them back with "
                               "ChainFile.to_chainlist()")
        return chain

    def get_param(self, param_idx, chain_idx=None):
        """The history of one parameter at one temperature: [iters] for a single ladder, [iters, R] for R replicas."""
        samples = self._memory_chain(chain_idx).get_samples()[:, :, param_idx]
        return samples[:, 0] if self.num_replicas == 1 else samples

    def get_sample(self, sample_idx, chain_idx=None):
        """The state saved at one iteration at one temperature: [P] for a single ladder, [R, P] for R replicas."""
        sample = self._memory_chain(chain_idx).get_samples()[sample_idx]
        return sample[0] if self.num_replicas == 1 else sample

    def set_current(self, theta, data=None):
        """Put every temperature of every replica at ``theta`` ([P] or [R, P]) and evaluate it there
        (multi_chain_serial_sampler.py:28-31)."""
        th = theta.detach().to(device=self.device, dtype=self.dtype)
        th = th[None] if th.dim() == 1 else th
        if th.shape[0] != self.num_replicas:
            th = th.expand(self.num_replicas, -1)
        self.sampler.set_current(th.repeat(self.num_chains, 1).contiguous(), data=data)

    def set_all(self, theta, data=None):
        # (the reference's own method calls itself, multi_chain_serial_sampler.py:33-36; what it means is this)
        self.set_current(theta, data=data)

    def reset_chains(self):
        for i, chain in enumerate(self.chains):
            if self.storage == 'list':
                chain.reset()
            else:
                for handle in ([chain] if self.num_replicas == 1 else chain):
                    handle.reset(keys=self.keys)

    def to_chainfile(self, path=Path.cwd(), mode='a'):
        """One directory per temperature, ``sampler<i>`` zero-filled as the reference does (to the width of the number of
        temperatures, multi_chain_serial_sampler.py:44-46), with ``<key>.csv`` inside; R replicas get ``replica<r>``
        directories below it."""
        path = Path(path)
        for i in range(self.num_chains):
            folder = path / ('sampler' + str(i).zfill(self.num_chains))
            buf = self._memory_chain(i)
            if self.num_replicas == 1:
                buf.get_chain(0).to_chainfile(keys=self.keys, path=folder, mode=mode)
            else:
                width = len(str(self.num_replicas))
                for r in range(self.num_replicas):
                    buf.get_chain(r).to_chainfile(keys=self.keys, path=folder / f"replica{r + 1:0{width}}", mode=mode)

    def reset(self, theta, data=None, reset_counter=True, reset_chain=True):
        if reset_counter:
            self.counter.reset()
        if reset_chain:
            self.reset_chains()
        th = theta.detach().to(device=self.device, dtype=self.dtype)
        th = th[None] if th.dim() == 1 else th
        self.sampler.set_current(th.repeat(self.num_chains, 1).contiguous(), data=data)
