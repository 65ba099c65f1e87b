#!/usr/bin/env python3
"""A/B the MFMA kernel variants in ONE process, interleaved rounds (cdna_hip_programming.md rule 24).
usage: python tools/ab_variants.py [variants...]   e.g. 0 1 3 5     (+1024: the exact f32 products instead of bf16x3)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L  # noqa: E402
from eeyore_amd.datasets import synthetic  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402

variants = [int(v) for v in sys.argv[1:]] or [0, 1, 3]
dev = torch.device("cuda", 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
plan = Plan([4, 32, 32, 3], [1, 1, 1], [1, 1, 0], 1, torch.float32, dev)
plan.set_data(torch.tensor(xs, dtype=torch.float32, device=dev), torch.tensor(ys, dtype=torch.float32, device=dev))
plan.set_prior(torch.zeros(plan.P), torch.full((plan.P,), float(np.sqrt(3.0))))
C = int(os.environ.get("AB_CHAINS", "4096"))
step = float(os.environ.get("AB_STEP", "0.011"))
theta = 0.1 * plan.philox_normal(C, seed=0, it=0)
target, grad = plan.log_target_grad(theta)
out = dict(accepted=plan.empty(C, dtype=torch.uint8), rate=plan.empty(C), h_cur=plan.empty(C), h_prop=plan.empty(C))
it = 1
IPL = int(os.environ.get("AB_IPL", "1"))  # > 1: time ey_hmc_run launches of IPL iterations with moments attached (the bench's form)
if IPL > 1:
    from eeyore_amd.distributed import ChainStats
    stats = ChainStats(C, plan.P, dev)
    stats.attach(plan)


def draw(n=1):
    global it
    for _ in range(n):
        if IPL > 1:
            plan.hmc_run(theta, target, grad, step, 20, IPL, seed=1, it=it, out=out); it += IPL
        else:
            plan.hmc_step(theta, target, grad, step, 20, seed=1, it=it, out=out); it += 1


draw(10 if IPL == 1 else 2)
torch.cuda.synchronize()
times = {v: [] for v in variants}
for rnd in range(8):
    for v in variants:
        plan.set_variant(v & 1023)
        plan.f32_products = 'exact' if v & 1024 else 'bf16x3'
        draw()  # warm the variant
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5 if IPL == 1 else 2
        a.record()
        draw(n)
        b.record()
        torch.cuda.synchronize()
        times[v].append(a.elapsed_time(b) / (n * IPL))
plan.set_variant(0)
F = 1092690 * 20 * C
print(f"chains {C} step {step} acceptance {out['accepted'].float().mean().item():.3f} "
      f"checksum {theta.double().sum().item():.10e} {target.double().sum().item():.10e}")
for v in variants:
    t = np.array(times[v])
    print(f"variant {v} ({'exact' if v & 1024 else 'bf16x3'} products): median {np.median(t):.4f} ms  min {t.min():.4f} ms  -> {F / np.median(t) / 1e9:.1f} TFLOP/s "
          f"({100 * F / np.median(t) / 1e9 / 157.3:.1f}% of f32 MFMA peak)")
