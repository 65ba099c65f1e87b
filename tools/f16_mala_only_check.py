#!/usr/bin/env python3
"""Diagnostic: the MALA draw alone of a fused16 build against the oracle (for single-mode builds, -DF16_ONLY_MODE=3: the
value and the gradient at the current position come from the oracle).  EEYORE_AMD_LIB=... python tools/f16_mala_only_check.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.c_oracle import COracle
from eeyore_amd.plan import Plan
DEV = "cuda:0"
dims, acts, lik, N = [4, 10, 7, 3], [1, 1, 0], 1, 150
npdt, dt = np.float32, torch.float32
rng = np.random.default_rng(sum(dims) + N)
x = rng.standard_normal((N, dims[0])); y = np.eye(dims[-1])[rng.integers(0, dims[-1], N)]
P = sum((dims[l] + 1) * dims[l + 1] for l in range(len(dims) - 1))
mu, sigma = 0.1 * rng.standard_normal(P), 0.5 + rng.random(P)
t_ = lambda a: torch.tensor(np.asarray(a), dtype=dt, device=DEV).contiguous()
pl = Plan(dims, [1] * 3, acts, lik, dt, DEV); pl.f32_products = "exact"
pl.set_data(t_(x), t_(y)); pl.set_prior(torch.tensor(mu), torch.tensor(sigma))
co = COracle(dims, acts, lik, x, y, mu, sigma, dtype=np.float64, nthreads=4)
C = 11
th0 = (0.3 * rng.standard_normal((C, P))).astype(npdt)
ref = [co.log_target_grad(th0[c].astype(np.float64)) for c in range(C)]
t0 = np.array([r[0] for r in ref]); g0 = np.stack([r[1] for r in ref])
p0 = rng.standard_normal((C, P)).astype(npdt); u = rng.random(C).astype(npdt)
th, tv, gg = t_(th0).clone(), t_(t0), t_(g0)
out = pl.mala_step(th, tv, gg, 0.004, z=t_(p0), u=t_(u))
f8 = lambda a_: np.asarray(a_, dtype=np.float64).copy()
acc, lr = co.mala_draw(f8(th0), t0.copy(), g0.copy(), f8(p0), f8(u), 0.004)
err = np.abs(out["log_rate"].cpu().numpy() - lr) / np.maximum(1, np.abs(lr))
print("mala log_rate max relative error %.3e  %s" % (err.max(), "PASS" if err.max() < 5e-3 else "FAIL"))
