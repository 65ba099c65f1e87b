"""Sampler base classes behind the reference's names (``Sampler``, ``SerialSampler``, ``SingleChainSerialSampler``).

Semantics kept from the reference: ``run`` walks epochs x batches, calls ``draw(x, y, savestate)`` once per batch with
``savestate`` true from the first post-burn-in iteration on, and advances the counter
(eeyore/samplers/serial_sampler.py:35-52); ``benchmark`` repeats whole runs until enough succeed and writes
``runNN/{<key>.csv, runtime.txt}``, ``runNN/errors/errorNN.txt`` and ``run_counts.txt`` (:54-126).
What differs is what one ``draw`` advances: the reference holds one chain, a sampler here holds C chains as device
tensors ``[C, ...]`` and advances all of them with one fused HIP launch.  ``theta0`` of shape [P] keeps the reference's
single-chain view of the state (``current['sample']`` is [P], ``accepted`` a python int, the chain a ``ChainList``).
"""
from datetime import timedelta
from pathlib import Path
from timeit import default_timer as timer

import torch

from eeyore_amd.chains import ChainBuffer, ChainList
from eeyore_amd.datasets import DataCounter, batches


class Sampler:
    def draw(self, x, y, savestate=False):
        raise NotImplementedError

    def run(self, num_epochs, num_burnin_epochs, verbose=False, verbose_step=100):
        raise NotImplementedError


class _Progress:
    """Times blocks of ``every`` iterations and prints one line per block when verbose."""

    def __init__(self, enabled, every, counter):
        self.enabled, self.every, self.counter = enabled, every, counter
        wi, we = len(str(counter.num_iters)), len(str(counter.num_epochs))
        self.template = (f'Iteration {{:{wi}}} out of {counter.num_iters} '
                         f'(in epoch {{:{we}}} out of {counter.num_epochs}), duration {{}}')
        self.t0 = None

    def before(self, idx):
        if self.enabled and idx % self.every == 0:
            self.t0 = timer()

    def after(self, idx, epoch):
        if self.enabled and (idx + 1) % self.every == 0:
            print(self.template.format(idx + 1, epoch + 1, timedelta(seconds=timer() - self.t0)))


def _write_line(path, text):
    with open(path, 'w') as handle:
        handle.write(f"{text}\n")


class SerialSampler(Sampler):
    def __init__(self, counter):
        self.counter = counter

    def run(self, num_epochs, num_burnin_epochs, verbose=False, verbose_step=100):
        counter = self.counter
        counter.set_epoch_info(num_epochs, num_burnin_epochs)
        progress = _Progress(verbose, verbose_step, counter)
        for epoch in range(counter.num_epochs):
            for x, y in batches(self.dataloader):
                progress.before(counter.idx)
                self.draw(x, y, savestate=counter.idx >= counter.num_burnin_iters)
                progress.after(counter.idx, epoch)
                counter.increment_idx()

    def benchmark(self, num_chains, num_epochs, num_burnin_epochs, path, init=None, check_conditions=None,
                  verbose=False, verbose_step=100, print_acceptance=False, print_runtime=True):
        root = Path(path)
        width = len(str(num_chains))
        done = rejected = crashed = 0
        while done < num_chains:
            if verbose:
                print(f'Simulating chain {done + 1:{width}} out of {num_chains} ({rejected:{width}} failures due to not '
                      f'meeting conditions and {crashed:{width}} failures due to runtime error)...')
            run_dir = root / f'run{str(done + 1).zfill(width)}'
            run_dir.mkdir(parents=True, exist_ok=True)
            try:
                start = self.get_model().prior.sample() if init is None else init[done]
                self.reset(start.clone().detach(), data=None, reset_counter=True, reset_chain=True)
                t0 = timer()
                self.run(num_epochs=num_epochs, num_burnin_epochs=num_burnin_epochs, verbose=verbose,
                         verbose_step=verbose_step)
                runtime = timer() - t0
            except RuntimeError as error:
                err_dir = run_dir / 'errors'
                err_dir.mkdir(parents=True, exist_ok=True)
                _write_line(err_dir / f'error{str(crashed + 1).zfill(num_chains)}.txt', error)
                crashed += 1
                if verbose:
                    print('Failed due to runtime error\n')
                continue
            accepted_run = check_conditions is None or check_conditions(self.get_chain(), runtime)
            if accepted_run:
                self.get_chain().to_chainfile(path=run_dir, mode='w')
                _write_line(run_dir / 'runtime.txt', runtime)
                done += 1
            else:
                rejected += 1
            if verbose:
                notes = ['Succeeded' if accepted_run else 'Failed due to not meeting conditions']
                if print_acceptance:
                    notes.append(f'acceptance rate = {self.get_chain().acceptance_rate()}')
                if print_runtime:
                    notes.append(f'runtime = {timedelta(seconds=runtime)}')
                print('; '.join(notes) + '\n')
        with open(root / 'run_counts.txt', 'w') as handle:
            handle.write(f"{done},succesful\n{rejected},unmet_conditions\n{crashed},runtime_errors\n")


class SingleChainSerialSampler(SerialSampler):
    """State handling shared by HMC / MALA / MetropolisHastings."""

    fused_block = 256  # iterations per launch of the fused run loop (0 disables it)

    def __init__(self, counter):
        super().__init__(counter=counter)

    # -- the run loop (serial_sampler.py:35-52) with whole blocks of iterations inside one launch where nothing on the
    #    host has to look at the state in between
    def _can_fuse(self, verbose):
        return (self.fused_block > 0 and self.batched and self.rng == 'philox' and not verbose
                and self.counter.num_batches == 1 and isinstance(self.chain, ChainBuffer)
                and set(self.chain.keys) <= {'sample', 'target_val', 'accepted'})

    def run(self, num_epochs, num_burnin_epochs, verbose=False, verbose_step=100):
        """As SerialSampler.run; for C chains on the in-kernel random streams with a full batch, the iterations in
        which no tuner adapts run in blocks of ``fused_block`` per launch (ey_hmc_run / ey_mala_run / ey_mh_run) and
        are recorded straight into the chain buffer.  The chains are the same, bit for bit, as with one launch per
        iteration."""
        if not self._can_fuse(verbose):
            return super().run(num_epochs, num_burnin_epochs, verbose=verbose, verbose_step=verbose_step)
        counter = self.counter
        counter.set_epoch_info(num_epochs, num_burnin_epochs)
        x, y = next(iter(self.dataloader))
        if not hasattr(self.model._plan(x, y), 'hmc_run'):  # a plan without the block entry points
            return super().run(num_epochs, num_burnin_epochs, verbose=verbose, verbose_step=verbose_step)
        plan = self.model._plan(x, y)
        tuner = getattr(self, 'tuner', None)
        # Whatever an earlier run left attached goes first: a run that raised or was interrupted during burn-in would
        # otherwise leave the plan (cached on the model) with device pointers into a tuner that may be gone by now.
        plan.detach_da()
        # a per-chain dual-averaging tuner can run inside the fused kernels' epilogue (ey_plan_attach_da): burn-in is
        # then blocks of iterations per launch too
        in_kernel_da = (hasattr(tuner, 'attach') and plan.kernel in ('mfma32', 'fused16')
                        and counter.num_burnin_iters > counter.idx)
        if in_kernel_da:
            tuner.attach(plan, counter.num_burnin_iters - counter.idx, idx0=counter.idx)
            self.step = tuner.step
        try:
            while counter.idx < counter.num_iters:
                burning = counter.idx < counter.num_burnin_iters
                if in_kernel_da and not burning:
                    tuner.detach()
                    self.step, in_kernel_da = tuner.step, False
                if burning and tuner is not None and not in_kernel_da:
                    self.draw(x, y, savestate=False)  # the tuner looks at every iteration's acceptance rates
                    counter.increment_idx()
                    continue
                k = min(self.fused_block, (counter.num_burnin_iters if burning else counter.num_iters) - counter.idx)
                self._draw_block(x, y, k, savestate=not burning)
                for _ in range(k):
                    counter.increment_idx()
        finally:
            if in_kernel_da:  # a run that ends (or fails, or is interrupted) inside burn-in
                tuner.detach()
                self.step = tuner.step

    def _draw_block(self, x, y, k, savestate):
        plan = self.model._plan(x, y)
        rec = {}
        if savestate:
            views = self.chain.block(k, dict(sample=self._theta, target_val=self._target,
                                             accepted=torch.empty(self.num_chains, dtype=torch.uint8,
                                                                  device=self._theta.device)))
            rec = dict(samples=views.get('sample'), targets=views.get('target_val'),
                       accepted_rec=views.get('accepted'))
        if not savestate and k > 1 and plan._moments is not None and plan.kernel != 'mfma32':
            # attached chain moments (ChainStats.attach) outside the mfma32 kernel are accumulated from the recorded
            # samples of a block; a burn-in block records nothing, so it goes iteration by iteration
            for _ in range(k):
                out = self._run_block(plan, 1, {})
                self._iter += 1
        else:
            out = self._run_block(plan, k, rec)
            self._iter += k
        self._publish(out['accepted'])
        self.last = out
        if savestate:
            self.chain.commit(k)

    def _run_block(self, plan, k, rec):
        raise NotImplementedError

    def _configure(self, model, dataloader, theta0, chain, rng, seed, chain_offset, temperature):
        self.model = model
        self.dataloader = dataloader
        self.temperature = temperature  # per-chain temperatures [C] (tempering); None -> model.temperature
        self.batched = theta0 is not None and theta0.dim() == 2
        self.num_chains = theta0.shape[0] if self.batched else 1
        self.chain = chain if chain is not None else (ChainBuffer() if self.batched else ChainList())
        self.rng = rng or ('philox' if self.batched else 'torch')
        if self.rng not in ('philox', 'torch'):
            raise ValueError("rng must be 'philox' (in-kernel counter-based stream) or 'torch' (global torch generator)")
        self.seed, self.chain_offset = int(seed), int(chain_offset)
        self._iter = 0

    # -- pieces the concrete samplers use
    def _temp(self):
        return self.temperature if self.temperature is not None else self.model.temperature

    def set_temperature(self, temperature):
        """Give the chains new temperatures [C] without touching their states (a replica of a tempering ladder that
        exchanged its temperature LABEL with a neighbour, ``distributed.TemperingExchange``): the cached tempered
        log-target and gradient are rescaled by t_new / t_old, which is what re-evaluating them would give
        (bayesian_model.py:33-34,48-49 multiply both by the temperature)."""
        old = self._temp()
        self.temperature = temperature
        if getattr(self, '_target', None) is not None:
            kw = dict(device=self._target.device, dtype=self._target.dtype)
            # no temperature so far = temperature one (bayesian_model.py:33-34: `if self.temperature is not None`)
            t_old = torch.ones((), **kw) if old is None else torch.as_tensor(old, **kw)
            new = self._temp()   # as `old` was obtained: a sampler without its own falls back to the model's
            t_new = torch.ones((), **kw) if new is None else torch.as_tensor(new, **kw)
            ratio = (t_new / t_old).expand(self._target.shape[0]).contiguous()
            self._target *= ratio
            if getattr(self, '_grad', None) is not None:
                self._grad *= ratio[:, None]

    def _step_args(self):
        """(scalar step, per-chain step vector or None) as the C ABI takes them."""
        if torch.is_tensor(self.step) and self.step.dim() == 1:
            return 0.0, self.step
        return float(self.step), None

    def _state_tensor(self, theta):
        th = theta.detach().to(device=self.model.device, dtype=self.model.dtype)
        return (th if th.dim() == 2 else th.unsqueeze(0)).contiguous().clone()

    def _expose(self, t):
        return t if self.batched else t[0]

    def _draw_randoms(self, C, P):
        """(normals [C, P], uniforms [C]) from the global torch generator, or (None, None) for the in-kernel stream."""
        if self.rng != 'torch':
            return None, None
        return self._randn(C, P), self._rand(C)

    def _randn(self, C, P):
        kw = dict(dtype=self.model.dtype, device=self.model.device)
        return torch.randn(C, P, **kw) if self.batched else torch.randn(P, **kw).unsqueeze(0)

    def _rand(self, C):
        return torch.rand(C if self.batched else 1, dtype=self.model.dtype, device=self.model.device)

    def _publish(self, accepted):
        cur = self.current
        cur['sample'] = self._expose(self._theta)
        cur['target_val'] = self._expose(self._target)
        if 'grad_val' in cur and hasattr(self, '_grad'):
            cur['grad_val'] = self._expose(self._grad)
        cur['accepted'] = accepted if self.batched else int(accepted[0].item())
        if not self.batched:
            self.model.set_params(cur['sample'])  # the model's parameters follow the chain (hmc.py:149-155)

    def _finish_draw(self, out, savestate):
        self._iter += 1
        self._publish(out['accepted'])
        self.last = out
        if savestate:
            self.chain.detach_and_update(self.current)

    # -- the reference's surface (single_chain_serial_sampler.py:10-41)
    def get_model(self):
        return self.model

    def get_chain(self):
        return self.chain

    def get_param(self, idx):
        return self.get_chain().get_param(idx)

    def get_sample(self, idx):
        return self.get_chain().get_sample(idx)

    def set_current(self, theta, data=None):
        self.current = dict.fromkeys(self.keys)
        self.current['sample'] = theta
        x, y = data or next(iter(self.dataloader))
        return x, y

    def set_all(self, theta, data=None):
        self.set_current(theta, data=data)

    def reset(self, theta, data=None, reset_counter=True, reset_chain=True):
        if reset_counter:
            self.counter.reset()
        if reset_chain:
            self.chain.reset(keys=self.chain.vals.keys())
        self.set_all(theta, data=data)

    def to_chainfile(self, path=Path.cwd(), mode='a'):
        self.chain.to_chainfile(path=path, mode=mode)


def default_counter(counter, dataloader):
    return counter or DataCounter.from_dataloader(dataloader)
