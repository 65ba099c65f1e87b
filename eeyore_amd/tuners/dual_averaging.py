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
    """Dual averaging with one step size PER CHAIN: the recurrence of ``HMCDATuner`` applied
    elementwise to a [C] tensor on the device, fed with the per-chain acceptance rates a fused HMC step returns.  The
    number of leapfrog steps stays fixed (all chains share one launch), so only the step size adapts; the result goes
    to the kernels as their per-chain ``step_vec``."""

    gamma, t0, kappa = 0.05, 10, 0.75

    def __init__(self, e0, num_steps, d=0.65, eub=None):
        import torch
        self._torch = torch
        self.e0 = e0.clone()
        self.fixed_num_steps = int(num_steps)
        self.d = d
        self.m = torch.log(10 * e0)
        self.logeub = None if eub is None else math.log(eub)
        self.logbare = torch.zeros_like(e0)
        self.barh = torch.zeros_like(e0)

    def num_steps(self, e=None):
        return self.fixed_num_steps

    def tune(self, rate, idx, return_e=True):
        """``rate`` [C]: this iteration's acceptance rates.  Returns (step [C], num_steps)."""
        torch = self._torch
        it = idx + 1
        w = 1.0 / (it + self.t0)
        rate = torch.nan_to_num(rate.to(self.barh.dtype), nan=0.0)
        self.barh += w * ((self.d - rate) - self.barh)
        loge = self.m - math.sqrt(it) / self.gamma * self.barh
        if self.logeub is not None:
            loge = torch.clamp(loge, max=self.logeub)
        self.logbare += it ** (-self.kappa) * (loge - self.logbare)
        return torch.exp(loge if return_e else self.logbare), self.fixed_num_steps
