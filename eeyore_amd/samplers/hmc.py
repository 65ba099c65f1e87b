import torch

from .single_chain_serial_sampler import SingleChainSerialSampler
from eeyore_amd.datasets import DataCounter
from eeyore_amd.tuners import HMCDATuner
from eeyore_amd import _lib as L


class HMC(SingleChainSerialSampler):
    """Hamiltonian Monte Carlo with the reference's constructor and surface (eeyore/samplers/hmc.py:8-170).

    One ``draw`` is one call of ``ey_hmc_step``: momentum draw, L leapfrog steps, Hamiltonians, accept and state
    update for every chain, fused in a HIP kernel.  Extensions over the reference: ``theta0`` may be [C, P]
    (C chains advanced together); ``rng`` ('torch' = momentum/uniform from the global torch generator as
    hmc.py:134,148 do, 'philox' = in-kernel counter-based stream keyed by (seed, chain, iteration));
    ``recompute_initial_grad=True`` re-evaluates the gradient at the start of each trajectory exactly as
    hmc.py:104 does (same values, one more evaluation).  ``chain`` defaults to a fresh chain per sampler (the
    reference's default argument is one ChainList shared by every sampler, hmc.py:11)."""

    def __init__(self, model, theta0=None, dataloader=None, data0=None, counter=None, step=0.1, num_steps=10,
                 tuner=None, chain=None, rng=None, seed=0, chain_offset=0, recompute_initial_grad=False,
                 temperature=None):
        super(HMC, self).__init__(counter or DataCounter.from_dataloader(dataloader))
        self.model = model
        self.temperature = temperature  # per-chain temperatures [C] (parallel tempering); None -> model.temperature
        self.dataloader = dataloader
        self.tuner = tuner
        self.recompute_initial_grad = recompute_initial_grad
        self._init_mode(theta0, chain, rng, seed, chain_offset)
        self.keys = ['sample', 'target_val', 'grad_val', 'momentum', 'hamiltonian', 'accepted']
        self._iter = 0

        if self.tuner is not None:
            if isinstance(self.tuner, HMCDATuner):
                if self.tuner.e0 is None:
                    self.init_step(theta0.clone().detach())
                    if self.tuner.eub is not None:
                        self.step = min(self.tuner.eub, self.step)
                    self.tuner.set_m(self.step)
                else:
                    self.step = self.tuner.e0
                self.num_steps = self.tuner.num_steps(self.step)
        else:
            self.step = step
            self.num_steps = num_steps

        if theta0 is not None:
            self.set_current(theta0.clone().detach(), data=data0)

    # ---- reference helpers (hmc.py:84-98)
    def potential_energy(self, position, x, y):
        return -self.model.log_target(position, x, y)

    def upto_grad_potential_energy(self, position, x, y):
        target_val, grad_val = self.model.upto_grad_log_target(position, x, y)
        return -target_val, -grad_val

    def log_proposal(self, momentum):
        return - 0.5 * torch.sum(momentum**2, dim=-1)

    def kinetic_energy(self, momentum):
        return -self.log_proposal(momentum)

    def hamiltonian(self, potential, momentum):
        return potential + self.kinetic_energy(momentum)

    def set_current(self, theta, data=None):
        x, y = super().set_current(theta, data=data)
        self._theta = self._state_tensor(theta)
        plan = self.model._plan(x, y)
        self._target, self._grad = plan.log_target_grad(self._theta, temp=self._temp())
        self._publish(torch.zeros(self.num_chains, dtype=torch.uint8))
        self.current['accepted'] = None

    def leapfrog(self, position0, momentum0, x, y):
        """hmc.py:100-124: (position_L, momentum_L, target_val, grad_val); L+1 gradient evaluations."""
        plan = self.model._plan(x, y)
        single = position0.dim() == 1
        th = (position0[None] if single else position0).detach().to(self.model.device, self.model.dtype).contiguous().clone()
        p = (momentum0[None] if single else momentum0).detach().to(self.model.device, self.model.dtype).contiguous().clone()
        step, step_vec = self._step_args()
        t, g = plan.leapfrog(th, p, step, self.num_steps, step_vec=step_vec, temp=self._temp())
        return (th[0], p[0], t[0], g[0]) if single else (th, p, t, g)

    def init_step(self, theta):
        """Step-size doubling/halving heuristic for the dual-averaging tuner (hmc.py:38-77), chain 0 only."""
        x, y = next(iter(self.dataloader))
        self.step = 1.
        self.num_steps = 1
        th = theta if theta.dim() == 1 else theta[0]
        th = th.to(self.model.device, self.model.dtype)
        mom = torch.randn(self.model.num_params(), dtype=self.model.dtype, device=self.model.device)
        cur_h = self.hamiltonian(-self.model.log_target(th.clone(), x, y), mom)
        _, pm, pt, _ = self.leapfrog(th, mom, x, y)
        ratio = torch.exp(cur_h - self.hamiltonian(-pt, pm))
        a = 2 * (ratio > 0.5) - 1
        while torch.pow(ratio, a) > torch.pow(2., -a):
            self.step = (torch.pow(2., a) * self.step).item()
            _, pm, pt, _ = self.leapfrog(th, mom, x, y)
            ratio = torch.exp(cur_h - self.hamiltonian(-pt, pm))

    def draw(self, x, y, savestate=False):
        """hmc.py:126-170."""
        plan = self.model._plan(x, y)
        C, P = self._theta.shape
        temp = self._temp()
        if self.counter.num_batches != 1:
            # minibatching: the cached target/gradient belong to another batch (hmc.py:129-131)
            self._target, self._grad = plan.log_target_grad(self._theta, temp=temp)
        p0 = u = None
        if self.rng == 'torch':
            p0 = self._randn(C, P)
            u = self._rand(C)
        flags = L.EY_RECOMPUTE_INITIAL_GRAD if self.recompute_initial_grad else 0
        step, step_vec = self._step_args()
        out = plan.hmc_step(self._theta, self._target, self._grad, step, self.num_steps, p0=p0, u=u, temp=temp,
                            step_vec=step_vec,
                            seed=self.seed, it=self._iter, chain_offset=self.chain_offset, flags=flags)
        self._iter += 1
        self._publish(out['accepted'])
        self.current['momentum'] = None if p0 is None else self._expose(p0)
        self.current['hamiltonian'] = self._expose(out['h_cur'])
        self.last = out

        if self.tuner is not None and isinstance(self.tuner, HMCDATuner):
            if self.counter.idx < self.counter.num_burnin_iters:
                rate = out['rate'].mean().item()
                self.step, self.num_steps = self.tuner.tune(
                    rate, self.counter.idx, return_e=self.counter.idx != self.counter.num_burnin_iters - 1)

        if savestate:
            self.chain.detach_and_update(self.current)
