#!/usr/bin/env python3
"""Measure the f32 error of the MALA / MH log-rates (and HMC rates) of the HIP kernels against an f64 evaluation of
the same f32 inputs (the C oracle in f64), next to the error of the C oracle in f32: the numbers the accept-decision
margins of tests/test_gpu_parity.py are set from."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eeyore_amd import _lib as L  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402
from oracle.c_oracle import COracle  # noqa: E402
from tests.helpers import groups, load  # noqa: E402

DEV = "cuda:0"
rec = dict(groups(load("g4_hmc_traces.npz"))["mlp432323_synth"])
dims, acts, lik = rec["dims"].tolist(), rec["acts"].tolist(), int(rec["lik"])


def oracle(dt):
    return COracle(dims, acts, lik, rec["x"], rec["y"], rec["prior_mu"], rec["prior_sigma"], dtype=dt, nthreads=8)


pl = Plan(dims, [1, 1, 1], acts, lik, torch.float32, DEV)
pl.set_data(torch.tensor(rec["x"], dtype=torch.float32, device=DEV), torch.tensor(rec["y"], dtype=torch.float32, device=DEV))
pl.set_prior(torch.tensor(rec["prior_mu"]), torch.tensor(rec["prior_sigma"]))
o32, o64 = oracle(np.float32), oracle(np.float64)
C, P = 512, pl.P
for scale0 in (0.2, 1.0):
    th0 = scale0 * pl.philox_normal(C, seed=4, it=0)
    t0, g0 = pl.log_target_grad(th0)
    z, u = pl.philox_normal(C, seed=4, it=1), pl.philox_uniform(C, seed=4, it=1)
    zn, un = z.cpu().numpy(), u.cpu().numpy()

    def truth(kind, par):
        th = th0.cpu().numpy().astype(np.float64); tv = t0.cpu().numpy().astype(np.float64); g = g0.cpu().numpy().astype(np.float64)
        if kind == "mala":
            return o64.mala_draw(th, tv, g, zn.astype(np.float64), un.astype(np.float64), par)[1]
        return o64.mh_draw(th, tv, zn.astype(np.float64), un.astype(np.float64), par)[1]

    def c32(kind, par):
        th = th0.cpu().numpy().copy(); tv = t0.cpu().numpy().copy(); g = g0.cpu().numpy().copy()
        if kind == "mala":
            return o32.mala_draw(th, tv, g, zn.copy(), un.copy(), par)[1]
        return o32.mh_draw(th, tv, zn.copy(), un.copy(), par)[1]

    for kind, pars in (("mala", (2e-4, 4e-3)), ("mh", (4e-3, 2e-2))):
        for par in pars:
            for flags, name in ((0, "mfma32"), (L.EY_FORCE_GENERIC, "generic")):
                a = [th0.clone(), t0.clone(), g0.clone()]
                if kind == "mala":
                    out = pl.mala_step(*a, par, z=z, u=u, flags=flags)
                else:
                    out = pl.mh_step(a[0], a[1], torch.full((P,), par), z=z, u=u, flags=flags)
                lr = out["log_rate"].cpu().numpy().astype(np.float64)
                tr = truth(kind, par)
                e = np.abs(lr - tr)
                e32 = np.abs(c32(kind, par).astype(np.float64) - tr)
                print(f"theta0 scale {scale0} {kind} par {par} {name}: |lr| median {np.median(np.abs(tr)):.3g}  "
                      f"gpu err max {e.max():.3e} p99 {np.quantile(e, 0.99):.3e} | C-f32 err max {e32.max():.3e} "
                      f"| in-margin(1e-2) {int((np.abs(np.log(un) - tr) < 1e-2).sum())} of {C}")
