import numpy as np

from .tuner import Tuner


class HMCDATuner(Tuner):
    """Dual-averaging step-size tuner, Hoffman & Gelman (2014) algorithms 4-5, with the reference's defaults and
    its optional upper bound ``eub`` (eeyore/tuners/hmcda_tuner.py:8-59).  Host-side scalar recurrence."""

    def __init__(self, l, e0=None, d=0.65, eub=None):
        self.l = l
        self.e0 = e0
        self.d = d
        self.eub = eub
        self.m = None if e0 is None else np.log(10 * e0)
        self.logeub = None if eub is None else np.log(eub)
        self.logbare = 0.
        self.barh = 0.
        self.g = 0.05
        self.t0 = 10
        self.k = 0.75

    def set_m(self, e0):
        self.m = np.log(10 * e0)

    def num_steps(self, e):
        return max(1, round(self.l / e))

    def tune(self, rate, idx, return_e=True):
        it = idx + 1
        d_w = 1 / (it + self.t0)
        e_w = 1 / (it ** self.k)
        self.barh = (1 - d_w) * self.barh + d_w * (self.d - rate)
        loge = self.m - np.sqrt(it) * self.barh / self.g
        if self.logeub is not None:
            loge = min(loge, self.logeub)
        self.logbare = e_w * loge + (1 - e_w) * self.logbare
        e = np.exp(loge) if return_e else np.exp(self.logbare)
        return e, self.num_steps(e)
