#!/usr/bin/env python3
"""f32-equivalence evidence for the bf16x3 form of the fused trajectory kernel (run on the GPU box):

    python tests/tools/bf3_error_probe.py > profiles/r03_bf3_error_probe.txt

For every f32 fixture of G2 (upto_grad_log_target) and G3 (HMC.leapfrog) on the headline model, and for 512 seeded chains
at two parameter scales, the exact kernel (v_mfma_f32_32x32x2_f32 products) and the bf16x3 kernel are compared with the
C oracle evaluated in f64 ON THE SAME f32 INPUTS: max and rms error of the log-target, the gradient, and the end point of
a trajectory.  The last column is the ratio bf16x3 / exact of the maximum errors (the bar: <= 1.5)."""
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


def plan_for(rec):
    dims, acts, lik = rec["dims"].tolist(), rec["acts"].tolist(), int(rec["lik"])
    pl = Plan(dims, [1] * (len(dims) - 1), acts, lik, torch.float32, DEV)
    pl.set_data(torch.tensor(rec["x"], dtype=torch.float32, device=DEV), torch.tensor(rec["y"], dtype=torch.float32, device=DEV))
    pl.set_prior(torch.tensor(rec["prior_mu"]), torch.tensor(rec["prior_sigma"]))
    temp = None if ("temperature" not in rec or np.isnan(rec["temperature"])) else float(rec["temperature"])
    o64 = COracle(dims, acts, lik, rec["x"].astype(np.float32).astype(np.float64), rec["y"], rec["prior_mu"],
                  np.asarray(rec["prior_sigma"], np.float32).astype(np.float64), dtype=np.float64, nthreads=8, temperature=temp)
    return pl, o64, temp


def both(pl, fn):
    out = {}
    for name in ("exact", "bf16x3"):
        pl.f32_products = name  # EY_OPT_F32_PRODUCTS of this plan
        out[name] = fn()
    return out


RMS_RATIOS, FIXTURE_RATIOS = [], []


def report(label, errs):
    """errs: {kernel: (abs error array, scale array)}"""
    row = []
    for k in ("exact", "bf16x3"):
        e, sc = errs[k]
        r = e / sc
        row.append((r.max(), np.sqrt((r ** 2).mean())))
    ratio = row[1][0] / max(row[0][0], 1e-300)
    RMS_RATIOS.append(row[1][1] / max(row[0][1], 1e-300))
    if label.startswith("G"):
        FIXTURE_RATIOS.append(ratio)
    print(f"{label:58s} exact max {row[0][0]:.3e} rms {row[0][1]:.3e} | bf16x3 max {row[1][0]:.3e} rms {row[1][1]:.3e} "
          f"| max ratio {ratio:.2f}")
    return ratio


def main():
    worst = 0.0
    print("errors are relative to the size of the quantity: |kernel - f64| / max(1, |f64|) for log-targets, and for vectors\n"
          "(gradient, theta_L, p_L of one chain) |kernel - f64| / max_i |f64_i| of that chain's vector")
    # ---- G2: value and gradient at the fixture's theta
    for name, rec in groups(load("g2_grads.npz")).items():
        if not (name.startswith("f32/") and "mlp432323" in name):
            continue
        pl, o64, temp = plan_for(rec)
        th = torch.tensor(rec["theta"], dtype=torch.float32, device=DEV)
        if th.dim() == 1:
            th = th[None]
        th = th.contiguous()
        C = th.shape[0]
        tt = np.zeros(C)
        gg = np.zeros((C, pl.P))
        for c in range(C):
            tt[c], gg[c], _, _ = o64.log_target_grad(th[c].cpu().numpy().astype(np.float64))
        res = both(pl, lambda: tuple(a.cpu().numpy().astype(np.float64) for a in pl.log_target_grad(th, temp=temp)))
        worst = max(worst, report(f"G2 {name} log-target", {k: (np.abs(v[0] - tt), np.maximum(1.0, np.abs(tt))) for k, v in res.items()}))
        worst = max(worst, report(f"G2 {name} gradient", {k: (np.abs(v[1] - gg), np.abs(gg).max(1, keepdims=True) + 0 * gg)
                                                          for k, v in res.items()}))
    # ---- G3: a whole trajectory from the fixture's start
    for name, rec in groups(load("g3_leapfrog.npz")).items():
        if not (name.startswith("f32/") and "mlp432323" in name):
            continue
        pl, o64, temp = plan_for(rec)
        th0 = np.asarray(rec["theta0"], np.float32)
        p0 = np.asarray(rec["p0"], np.float32)
        step, Ls = float(rec["step"]), int(rec["L"])
        thL, pL, tL, gL = o64.leapfrog(th0.astype(np.float64), p0.astype(np.float64), float(np.float32(step)), Ls)

        def run():
            th = torch.tensor(th0, device=DEV)[None].clone()
            p = torch.tensor(p0, device=DEV)[None].clone()
            t, g = pl.leapfrog(th, p, step, Ls)
            return th[0].cpu().numpy().astype(np.float64), p[0].cpu().numpy().astype(np.float64), t.item(), g[0].cpu().numpy().astype(np.float64)

        res = both(pl, run)
        worst = max(worst, report(f"G3 {name} theta_L", {k: (np.abs(v[0] - thL), np.abs(thL).max() + 0 * thL) for k, v in res.items()}))
        worst = max(worst, report(f"G3 {name} p_L", {k: (np.abs(v[1] - pL), np.abs(pL).max() + 0 * pL) for k, v in res.items()}))
        worst = max(worst, report(f"G3 {name} gradient", {k: (np.abs(v[3] - gL), np.abs(gL).max() + 0 * gL) for k, v in res.items()}))
    # ---- 512 seeded chains at two scales: value, gradient
    rec = dict(groups(load("g4_hmc_traces.npz"))["mlp432323_synth"])
    pl, o64, temp = plan_for(rec)
    for scale0 in (0.1, 1.0, 3.0):
        C = 512
        th = (scale0 * pl.philox_normal(C, seed=11, it=0)).contiguous()
        thn = th.cpu().numpy().astype(np.float64)
        tt = np.zeros(C)
        gg = np.zeros((C, pl.P))
        for c in range(C):
            tt[c], gg[c], _, _ = o64.log_target_grad(thn[c])
        res = both(pl, lambda: tuple(a.cpu().numpy().astype(np.float64) for a in pl.log_target_grad(th)))
        worst = max(worst, report(f"512 chains, theta ~ {scale0} N(0,1): log-target",
                                  {k: (np.abs(v[0] - tt), np.maximum(1.0, np.abs(tt))) for k, v in res.items()}))
        worst = max(worst, report(f"512 chains, theta ~ {scale0} N(0,1): gradient",
                                  {k: (np.abs(v[1] - gg), np.abs(gg).max(1, keepdims=True) + 0 * gg) for k, v in res.items()}))
    # ---- 256 seeded trajectories (L = 20, the benchmark's step): end points
    C, step, Ls = 256, 0.024, 20
    th0 = (0.3 * pl.philox_normal(C, seed=12, it=0)).contiguous()
    p0 = pl.philox_normal(C, seed=12, it=1).contiguous()
    thL, pL, gL = np.zeros((C, pl.P)), np.zeros((C, pl.P)), np.zeros((C, pl.P))
    for c in range(C):
        thL[c], pL[c], _, gL[c] = o64.leapfrog(th0[c].cpu().numpy().astype(np.float64), p0[c].cpu().numpy().astype(np.float64),
                                               float(np.float32(step)), Ls)

    def run_many():
        th, p = th0.clone(), p0.clone()
        t, g = pl.leapfrog(th, p, step, Ls)
        return tuple(a.cpu().numpy().astype(np.float64) for a in (th, p, g))

    res = both(pl, run_many)
    for i, (nm, tr) in enumerate((("theta_L", thL), ("p_L", pL), ("gradient", gL))):
        worst = max(worst, report(f"256 trajectories, L = 20, step 0.024: {nm}",
                                  {k: (np.abs(v[i] - tr), np.abs(tr).max(1, keepdims=True) + 0 * tr) for k, v in res.items()}))
    print(f"worst ratio of maximum errors, bf16x3 / exact: {worst:.2f}  (bar: <= 1.5); on the G2 / G3 fixtures alone "
          f"{max(FIXTURE_RATIOS):.2f}; worst ratio of rms errors {max(RMS_RATIOS):.2f}\n"
          "(the maxima over 256 chaotic trajectories of 20 steps are order statistics of a few hundred thousand elements: the\n"
          " rms columns are the stable comparison)")
    return 0 if worst <= 1.5 else 1


if __name__ == "__main__":
    sys.exit(main())
