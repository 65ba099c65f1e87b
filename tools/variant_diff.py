#!/usr/bin/env python3
"""Dump what one library build computes from a fixed state (one evaluation, one HMC draw, a 3-iteration run) so that
builds can be compared bit for bit:  EEYORE_AMD_LIB=tools/abl/lib_X.so python tools/variant_diff.py X
then                                  python tools/variant_diff.py --compare A B ..."""
import os
import sys

import numpy as np

OUT = "gpurun_out"
if sys.argv[1] == "--compare":
    ref = np.load(f"{OUT}/vd_{sys.argv[2]}.npz")
    for tag in sys.argv[3:]:
        z = np.load(f"{OUT}/vd_{tag}.npz")
        for k in ref.files:
            a, b = ref[k], z[k]
            same = np.array_equal(a, b, equal_nan=True)
            msg = "identical" if same else f"DIFFERENT: max |d| {np.nanmax(np.abs(a.astype(np.float64) - b)):.3e}, " \
                                           f"{int((a != b).sum())} of {a.size} elements"
            if not same and a.ndim == 2 and a.shape[1] == 1315:
                cols = np.unique(np.nonzero(a != b)[1])
                seg = [("W0", 0, 128), ("b0", 128, 160), ("W1", 160, 1184), ("b1", 1184, 1216), ("W2", 1216, 1312), ("b2", 1312, 1315)]
                msg += " in " + ",".join(f"{n}:{int(((cols >= lo) & (cols < hi)).sum())}" for n, lo, hi in seg)
            print(f"{sys.argv[2]} vs {tag}: {k:14s} {msg}")
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.datasets import synthetic  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402

tag = sys.argv[1]
dev = torch.device("cuda", 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
plan = Plan([4, 32, 32, 3], [1, 1, 1], [1, 1, 0], 1, torch.float32, dev)
plan.set_data(torch.tensor(xs, dtype=torch.float32, device=dev), torch.tensor(ys, dtype=torch.float32, device=dev))
plan.set_prior(torch.zeros(plan.P), torch.full((plan.P,), float(np.sqrt(3.0))))
C = 512
theta = 0.1 * plan.philox_normal(C, seed=0, it=0)
t0, g0 = plan.log_target_grad(theta)
res = dict(eval_target=t0.cpu().numpy(), eval_grad=g0.cpu().numpy())
th, tv, g = theta.clone(), t0.clone(), g0.clone()
out = plan.hmc_step(th, tv, g, 0.024, 20, seed=3, it=1)
res.update(step_theta=th.cpu().numpy(), step_target=tv.cpu().numpy(), step_grad=g.cpu().numpy(),
           step_rate=out["rate"].cpu().numpy(), step_hprop=out["h_prop"].cpu().numpy())
th, tv, g = theta.clone(), t0.clone(), g0.clone()
tl, gl = plan.leapfrog(th, plan.philox_normal(C, seed=5, it=2), 0.02, 1)
res.update(lf1_theta=th.cpu().numpy(), lf1_target=tl.cpu().numpy(), lf1_grad=gl.cpu().numpy())
th, tv, g = theta.clone(), t0.clone(), g0.clone()
plan.hmc_run(th, tv, g, 0.024, 20, 3, seed=3, it=1)
res.update(run_theta=th.cpu().numpy(), run_target=tv.cpu().numpy())
np.savez(f"{OUT}/vd_{tag}.npz", **res)
print("saved", tag, float(t0.double().sum()))
