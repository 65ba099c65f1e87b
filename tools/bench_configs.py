#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json (parity-test cases, not bench lines): throughput for the record.
cfg1: HMC, 1 chain, MLP(2-2-1), XOR, f64;  cfg2: MALA, 256 chains, MLP(2-3-2-1), binary synthetic N=256."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.datasets import XYDataset, synthetic  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


# cfg1
xor = XYDataset.from_eeyore('xor', dtype=torch.float64, device=dev)
pl = Plan([2, 2, 1], [1, 1], [1, 1], 0, torch.float64, dev)
pl.set_data(xor.x, xor.y)
pl.set_prior(torch.zeros(9), torch.full((9,), 100.0))
th = torch.tensor([[1.1, -2.9, -0.4, 0.8, 4.3, 9.2, 4.44, -3.4, 7.2]], dtype=torch.float64, device=dev)
t, g = pl.log_target_grad(th)
it = [0]
def f1():
    it[0] += 1
    pl.hmc_step(th, t, g, 0.1, 10, seed=1, it=it[0])
dt = timed(f1, 200)
print(f"cfg1 HMC 1 chain MLP(2-2-1) XOR f64 L=10: {dt * 1e6:.1f} us/draw -> {10 / dt:.3e} leapfrog-steps/s x chains "
      f"(launch-latency bound)")

th = torch.tensor([[1.1, -2.9, -0.4, 0.8, 4.3, 9.2, 4.44, -3.4, 7.2]], dtype=torch.float64, device=dev)
t, g = pl.log_target_grad(th)
K = 1000
def f1run():
    it[0] += K
    pl.hmc_run(th, t, g, 0.1, 10, K, seed=1, it=it[0])
dt = timed(f1run, 10) / K
print(f"cfg1 as above, 1000 iterations per launch (ey_hmc_run): {dt * 1e6:.2f} us/draw -> {10 / dt:.3e} "
      f"leapfrog-steps/s x chains")

# cfg2
for dtype in (torch.float32, torch.float64):
    data = synthetic.binary_xor_like(256, dtype=dtype, device=dev)
    pl = Plan([2, 3, 2, 1], [1, 1, 1], [1, 1, 1], 0, dtype, dev)
    pl.set_data(data.x, data.y)
    pl.set_prior(torch.zeros(20), torch.full((20,), float(np.sqrt(3.0))))
    for C in (256, 65536):
        th = 0.5 * pl.philox_normal(C, seed=0, it=0)
        t, g = pl.log_target_grad(th)
        accs = []
        def f2():
            it[0] += 1
            accs.append(pl.mala_step(th, t, g, 0.02, seed=2, it=it[0])["accepted"])
        dt = timed(f2, 100)
        acc = torch.stack(accs[-50:]).float().mean().item()
        print(f"cfg2 MALA {C} chains MLP(2-3-2-1) N=256 {str(dtype)[6:]}: {dt * 1e6:.1f} us/draw -> {C / dt:.3e} "
              f"draws/s x chains, acceptance {acc:.2f}")
        K2 = 100
        def f2run():
            it[0] += K2
            pl.mala_run(th, t, g, 0.02, K2, seed=2, it=it[0])
        dt = timed(f2run, 10) / K2
        print(f"     the same, {K2} iterations per launch (ey_mala_run): {dt * 1e6:.1f} us/draw -> {C / dt:.3e} "
              f"draws/s x chains")
