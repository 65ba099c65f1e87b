import torch

from .base import SingleChainSerialSampler, default_counter
from eeyore_amd.tuners import HMCDATuner, PerChainDATuner
from eeyore_amd import _lib as L


class HMC(SingleChainSerialSampler):
    """Hamiltonian Monte Carlo with the reference's constructor and surface (eeyore/samplers/hmc.py:8-170).

    One ``draw`` is one call of ``ey_hmc_step``: momentum draw, L leapfrog steps, Hamiltonians, accept and state
    update for every chain, fused in a HIP kernel.  Extensions over the reference: ``theta0`` may be [C, P]
    (C chains advanced together); ``rng`` ('torch' = momentum/uniform from the global torch generator as
    hmc.py:134,148 do, 'philox' = in-kernel counter-based stream keyed by (seed, chain, iteration));
    ``recompute_initial_grad=True`` re-evaluates the gradient at the start of each trajectory exactly as hmc.py:104
    does (same values, one more evaluation); ``step`` may be a [C] tensor and ``temperature`` a [C] tensor.
    ``chain`` defaults to a fresh chain per sampler (the reference's default argument is one ChainList shared by
    every sampler, hmc.py:11).  ``init_step_mode``: how a tuner without a starting step gets one -- 'intended'
    (default: the step-doubling heuristic hmc.py:38-77 set out to write) or 'reference' (what those lines compute,
    integer powers included; see ``init_step``)."""

    keys = ['sample', 'target_val', 'grad_val', 'momentum', 'hamiltonian', 'accepted']

    def __init__(self, model, theta0=None, dataloader=None, data0=None, counter=None, step=0.1, num_steps=10,
                 tuner=None, chain=None, rng=None, seed=0, chain_offset=0, recompute_initial_grad=False,
                 temperature=None, init_step_mode='intended'):
        super().__init__(default_counter(counter, dataloader))
        self._configure(model, dataloader, theta0, chain, rng, seed, chain_offset, temperature)
        if init_step_mode not in ('intended', 'reference'):
            raise ValueError("init_step_mode must be 'intended' or 'reference'")
        self.tuner = tuner
        self.recompute_initial_grad = recompute_initial_grad
        self.step, self.num_steps = step, num_steps
        if isinstance(tuner, HMCDATuner):
            if tuner.e0 is None:  # find a starting step size, then let dual averaging shrink towards it (hmc.py:17-27)
                self.init_step(theta0.clone().detach(), intended=init_step_mode == 'intended')
                if not self.step > 0:
                    # 'reference' mode, halving direction: the reference's 2**(-1) on an integer tensor is 0 and its
                    # tuner.num_steps then divides by zero (hmc.py:27,60); the same exception class, with the reason
                    raise ZeroDivisionError(
                        "HMC.init_step (init_step_mode='reference') collapsed the step to 0 as eeyore's integer "
                        "torch.pow(2, -1) does; pass init_step_mode='intended' or give the tuner a starting step e0")
                if tuner.eub is not None:
                    self.step = min(tuner.eub, self.step)
                tuner.set_m(self.step)
            else:
                self.step = tuner.e0
            self.num_steps = tuner.num_steps(self.step)
        elif isinstance(tuner, PerChainDATuner):
            self.step, self.num_steps = tuner.step, tuner.num_steps()
        if theta0 is not None:
            self.set_current(theta0.clone().detach(), data=data0)

    # -- energies (hmc.py:84-98); momentum may carry a leading chain axis
    def potential_energy(self, position, x, y):
        return -self.model.log_target(position, x, y)

    def upto_grad_potential_energy(self, position, x, y):
        target_val, grad_val = self.model.upto_grad_log_target(position, x, y)
        return -target_val, -grad_val

    def log_proposal(self, momentum):
        return -0.5 * momentum.pow(2).sum(-1)

    def kinetic_energy(self, momentum):
        return -self.log_proposal(momentum)

    def hamiltonian(self, potential, momentum):
        return potential + self.kinetic_energy(momentum)

    def set_current(self, theta, data=None):
        x, y = super().set_current(theta, data=data)
        self._theta = self._state_tensor(theta)
        self._target, self._grad = self.model._plan(x, y).log_target_grad(self._theta, temp=self._temp())
        self._publish(torch.zeros(self.num_chains, dtype=torch.uint8))
        self.current['accepted'] = None

    def leapfrog(self, position0, momentum0, x, y):
        """(position_L, momentum_L, target_val, grad_val) as hmc.py:100-124: L steps, L+1 gradient evaluations, the
        final momentum negated."""
        single = position0.dim() == 1
        th, p = self._state_tensor(position0), self._state_tensor(momentum0)
        step, step_vec = self._step_args()
        t, g = self.model._plan(x, y).leapfrog(th, p, step, self.num_steps, step_vec=step_vec, temp=self._temp())
        return (th[0], p[0], t[0], g[0]) if single else (th, p, t, g)

    def init_step(self, theta, intended=False):
        """First step size for the dual-averaging tuner, as ``HMC.init_step`` (hmc.py:38-77) computes it for chain 0:
        from step 1 with one leapfrog step per trial, the same momentum in every trial, r = exp(H_cur - H_prop).

        The reference's direction a = 2 (r > 1/2) - 1 is an integer tensor, so its ``torch.pow(2, -a)`` / ``torch.pow(2, a)``
        (:60, :67) are integer powers and 2**(-1) is 0: for a = +1 the step doubles until r underflows to 0 (or is NaN),
        for a = -1 it becomes 0 at once (``tuner.num_steps`` then raises ZeroDivisionError, :27).  ``intended=False``
        reproduces exactly that (pinned by the G9 fixture; the constructor's ``init_step_mode='reference'``);
        ``intended=True`` -- what the constructor uses unless told otherwise -- gives what :58-77 set out to write
        (Hoffman & Gelman 2014, algorithm 4: double or halve until r crosses 1/2), which is also what
        ``init_step_per_chain`` does.  (x, y) is the first batch of the loader and h_start is computed once, as in the
        full-batch case of the reference; with minibatches the reference draws a new batch per trial.)"""
        x, y = next(iter(self.dataloader))
        self.step, self.num_steps = 1., 1
        th = (theta if theta.dim() == 1 else theta[0]).to(self.model.device, self.model.dtype)
        momentum = torch.randn(self.model.num_params(), dtype=self.model.dtype, device=self.model.device)
        h_start = self.hamiltonian(-self.model.log_target(th.clone(), x, y), momentum)

        def ratio_after_one_step():
            _, p_new, t_new, _ = self.leapfrog(th, momentum, x, y)
            return torch.exp(h_start - self.hamiltonian(-t_new, p_new)).item()

        ratio = ratio_after_one_step()
        direction = 1 if ratio > 0.5 else -1
        if intended:
            threshold, factor = 2. ** (-direction), 2. ** direction
        else:
            threshold, factor = (0, 2) if direction == 1 else (2, 0)
        for _ in range(2200):  # f64 steps leave the representable range long before
            if not (ratio ** direction if ratio != 0 or direction == 1 else float('inf')) > threshold:
                break
            self.step = factor * self.step
            ratio = ratio_after_one_step()
        self.step = float(self.step)

    def init_step_per_chain(self, theta, max_trials=60):
        """One starting step size PER CHAIN for ``theta`` [C, P]: Hoffman & Gelman's heuristic
        (2014, algorithm 4) for every chain at once -- each trial is one launch of ``ey_hmc_leapfrog`` with the per-chain
        step vector; chains whose ratio has crossed 1/2 keep their step.  Returns steps [C] on the device."""
        x, y = next(iter(self.dataloader))
        plan = self.model._plan(x, y)
        th0 = self._state_tensor(theta)
        C, P = th0.shape
        momentum = torch.randn(C, P, dtype=self.model.dtype, device=self.model.device)
        t0, _ = plan.log_target_grad(th0, temp=self._temp())
        h_start = -t0 + 0.5 * momentum.pow(2).sum(-1)
        step = torch.ones(C, dtype=self.model.dtype, device=self.model.device)

        def ratio_after_one_step():
            th, p = th0.clone(), momentum.clone()
            t, _ = plan.leapfrog(th, p, 0.0, 1, step_vec=step, temp=self._temp())
            return torch.exp(h_start - (-t + 0.5 * p.pow(2).sum(-1)))

        ratio = torch.nan_to_num(ratio_after_one_step(), nan=0.0)
        direction = torch.where(ratio > 0.5, 1.0, -1.0).to(step.dtype)
        for _ in range(max_trials):
            active = ratio.pow(direction) > torch.pow(2.0, -direction)
            if not bool(active.any()):
                break
            step = torch.where(active, step * torch.pow(2.0, direction), step)
            ratio = torch.nan_to_num(ratio_after_one_step(), nan=0.0)
        return step

    def _run_block(self, plan, k, rec):
        step, step_vec = self._step_args()
        out = plan.hmc_run(self._theta, self._target, self._grad, step, self.num_steps, k, step_vec=step_vec,
                           temp=self._temp(), seed=self.seed, it=self._iter, chain_offset=self.chain_offset,
                           flags=L.EY_RECOMPUTE_INITIAL_GRAD if self.recompute_initial_grad else 0, **rec)
        self.current['momentum'] = None
        self.current['hamiltonian'] = None
        return out

    def draw(self, x, y, savestate=False):
        """One HMC iteration of every chain (hmc.py:126-170)."""
        plan = self.model._plan(x, y)
        temp = self._temp()
        if self.counter.num_batches != 1:
            # minibatching: the cached target/gradient belong to another batch (hmc.py:129-131)
            self._target, self._grad = plan.log_target_grad(self._theta, temp=temp)
        p0, u = self._draw_randoms(*self._theta.shape)
        step, step_vec = self._step_args()
        out = plan.hmc_step(self._theta, self._target, self._grad, step, self.num_steps, p0=p0, u=u, temp=temp,
                            step_vec=step_vec, seed=self.seed, it=self._iter, chain_offset=self.chain_offset,
                            flags=L.EY_RECOMPUTE_INITIAL_GRAD if self.recompute_initial_grad else 0)
        self._publish(out['accepted'])
        self.current['momentum'] = None if p0 is None else self._expose(p0)
        self.current['hamiltonian'] = self._expose(out['h_cur'])
        if isinstance(self.tuner, HMCDATuner) and self.counter.idx < self.counter.num_burnin_iters:
            last_burnin = self.counter.idx == self.counter.num_burnin_iters - 1
            self.step, self.num_steps = self.tuner.tune(out['rate'].mean().item(), self.counter.idx,
                                                        return_e=not last_burnin)
        elif isinstance(self.tuner, PerChainDATuner) and self.counter.idx < self.counter.num_burnin_iters:
            last_burnin = self.counter.idx == self.counter.num_burnin_iters - 1
            self.step, _ = self.tuner.tune(out['rate'], self.counter.idx, return_e=not last_burnin)
        self._iter += 1
        self.last = out
        if savestate:
            self.chain.detach_and_update(self.current)
