#!/usr/bin/env python3
"""Diagnostic: which part of a wrong MALA log-rate of a fused16 build is wrong?  log_rate = (tv - t_old) + (qf - qb) / (2 step).
Runs the failing case, recomputes the proposal on the host, takes the kernel's own value at the proposal from a value call and
from the accepted chains' stored targets, and prints the parts beside the oracle's.   EEYORE_AMD_LIB=... python tools/f16_mala_parts.py"""
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
t, g = pl.log_target_grad(t_(th0))
p0 = rng.standard_normal((C, P)).astype(npdt); u = np.full(C, 1e-30, npdt)   # accept everything: the stored target is tv
step = 0.004
th, tv, gg = t_(th0).clone(), t.clone(), g.clone()
out = pl.mala_step(th, tv, gg, step, z=t_(p0), u=t_(u))
lr = out["log_rate"].cpu().numpy().astype(np.float64)
# host: the proposal and the two quadratic forms in f64 from the kernel's own t, g
g0 = g.cpu().numpy().astype(np.float64); t0 = t.cpu().numpy().astype(np.float64)
prop = th0.astype(np.float64) + 0.5 * step * g0 + np.sqrt(step) * p0.astype(np.float64)
tp, gp = pl.log_target_grad(t_(prop))
tp = tp.cpu().numpy().astype(np.float64); gp = gp.cpu().numpy().astype(np.float64)
qf = ((prop - (th0 + 0.5 * step * g0)) ** 2).sum(1); qb = ((th0 - (prop + 0.5 * step * gp)) ** 2).sum(1)
want = (tp - t0) + (qf - qb) / (2 * step)
print("kernel log_rate          ", np.round(lr, 3))
print("host from kernel's parts ", np.round(want, 3))
print("stored target - tp       ", np.round(tv.cpu().numpy() - tp, 4), " accepted", out["accepted"].cpu().numpy())
print("new theta - proposal max ", np.abs(th.cpu().numpy() - prop).max(1).round(6))
print("new grad - gp max        ", np.abs(gg.cpu().numpy() - gp).max(1).round(5))
print("(lr - (tv - t0)) * 2 step", np.round((lr - (tv.cpu().numpy() - t0)) * 2 * step, 5), " want qf - qb", np.round(qf - qb, 5))
