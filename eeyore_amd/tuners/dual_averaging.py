"""Step-size adaptation by dual averaging (Hoffman & Gelman 2014, algorithms 4 and 5) behind the reference's names
``Tuner`` / ``HMCDATuner`` with its constructor, defaults and optional upper bound ``eub``
(eeyore/tuners/hmcda_tuner.py:8-59).  ``l`` is the trajectory length: ``num_steps(e) = max(1, round(l / e))``."""
import math


class Tuner:
    def tune(self, *args, **kwargs):
        raise NotImplementedError


class HMCDATuner(Tuner):
    gamma, t0, kappa = 0.05, 10, 0.75  # shrinkage, early-iteration damping, averaging decay (the paper's defaults)

    def __init__(self, l, e0=None, d=0.65, eub=None):
        self.l, self.e0, self.d, self.eub = l, e0, d, eub
        self.m = None if e0 is None else self._log10x(e0)         # mu: the point log-steps shrink towards
        self.logeub = None if eub is None else math.log(eub)
        self.logbare = 0.                                         # log of the averaged step
        self.barh = 0.                                            # running mean of (target - observed) acceptance
        self.g, self.k = self.gamma, self.kappa                   # the reference's attribute names

    @staticmethod
    def _log10x(e0):
        # numpy's log(0) = -inf (hmcda_tuner.py:32 warns and goes on; num_steps then divides by zero, hmc.py:27)
        return math.log(10 * e0) if e0 > 0 else float('-inf')

    def set_m(self, e0):
        self.m = self._log10x(e0)

    def num_steps(self, e):
        return max(1, round(self.l / e))

    def tune(self, rate, idx, return_e=True):
        """Feed the acceptance rate of iteration ``idx``; returns (step, num_steps): the exploring step while adapting,
        the averaged one when ``return_e`` is false (the last burn-in iteration)."""
        it = idx + 1
        w = 1.0 / (it + self.t0)
        self.barh += w * ((self.d - rate) - self.barh)
        loge = self.m - math.sqrt(it) / self.g * self.barh
        if self.logeub is not None:
            loge = min(loge, self.logeub)
        eta = it ** (-self.k)
        self.logbare += eta * (loge - self.logbare)
        e = math.exp(loge if return_e else self.logbare)
        return e, self.num_steps(e)


class PerChainDATuner(Tuner):
    """Dual averaging with one step size PER CHAIN: the recurrence of ``HMCDATuner`` applied to
    every chain on its own, fed with the per-chain acceptance rates of the fused HMC step.  The number of leapfrog steps
    stays fixed (all chains share one launch), so only the step size adapts; the result goes to the kernels as their
    per-chain ``step_vec``.

    Two ways to run it, with the same arithmetic (float64 state, the per-iteration coefficients worked out once on the
    host): ``tune`` after every iteration on the host side of the launch, or ``attach(plan, n)`` -- the recurrence then
    runs inside the step kernels' epilogue (``ey_plan_attach_da``), so a whole burn-in is blocks of iterations per
    launch with nothing read back in between."""

    gamma, t0, kappa = 0.05, 10, 0.75

    def __init__(self, e0, num_steps, d=0.65, eub=None):
        import torch
        self._torch = torch
        self.e0 = e0.clone()
        self.fixed_num_steps = int(num_steps)
        self.d = d
        self.eub = eub
        self.logeub = None if eub is None else math.log(eub)
        self.m = torch.log(10 * e0.to(torch.float64))          # mu: the point log-steps shrink towards
        self.logbare = torch.zeros_like(self.m)                 # log of the averaged step
        self.barh = torch.zeros_like(self.m)                    # running mean of (target - observed) acceptance
        self.step = e0.clone()                                  # the step every chain takes next
        self._attached = None

    def num_steps(self, e=None):
        return self.fixed_num_steps

    def coefficients(self, idx):
        """(1/(t + t0), sqrt(t)/gamma, t^-kappa) of the update after iteration ``idx`` (t = idx + 1)."""
        it = idx + 1
        return 1.0 / (it + self.t0), math.sqrt(it) / self.gamma, it ** (-self.kappa)

    def tune(self, rate, idx, return_e=True):
        """``rate`` [C]: this iteration's acceptance rates.  Returns (step [C], num_steps)."""
        torch = self._torch
        w, sq, ew = self.coefficients(idx)
        rate = torch.nan_to_num(rate.to(torch.float64), nan=0.0)
        self.barh += w * ((self.d - rate) - self.barh)
        loge = self.m - sq * self.barh
        if self.logeub is not None:
            loge = torch.clamp(loge, max=self.logeub)
        self.logbare += ew * (loge - self.logbare)
        self.step = torch.exp(loge if return_e else self.logbare).to(self.e0.dtype)
        return self.step, self.fixed_num_steps

    # -- the same recurrence inside the step kernels
    def attach(self, plan, n, idx0=0, final_avg=True):
        """From now on the plan's HMC launches adapt ``self.step`` themselves for the next ``n`` iterations (iteration
        indices idx0 .. idx0 + n - 1); with ``final_avg`` the last of them leaves the averaged step, as the reference
        asks of its tuner at the end of burn-in (hmc.py:158-163)."""
        import ctypes as ct
        from eeyore_amd import _lib as L
        torch = self._torch
        dev = self.step.device
        self._state = torch.stack([self.barh, self.logbare, self.m], dim=1).to(dev).contiguous()   # [C, 3] float64
        self._table = torch.tensor([self.coefficients(idx0 + k) for k in range(n)], dtype=torch.float64,
                                   device=dev).contiguous()
        self.step = self.step.contiguous()
        plan.attach_da(self._state, self.step, self._table, n, self.d, self.logeub, final_avg)
        self._attached = plan

    def detach(self):
        """Stop adapting inside the kernels and take their state back (``barh``, ``logbare``; ``step`` was theirs)."""
        if self._attached is None:
            return
        self._attached.detach_da()
        self.barh, self.logbare = self._state[:, 0].clone(), self._state[:, 1].clone()
        self._attached = None
