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
    pl.row_waves = "auto"   # the opt-in latency setting (the default, 'off', keeps a chain's bits independent of the chain count)
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

# ---- fused 16x16x4 kernels (ey_fused16.hip): the headline model in f64 and other widths / likelihoods in f32,
# HMC L = 20 on Iris-shaped synthetic data, against the generic VALU kernel on the same plan (EY_FORCE_GENERIC)
from eeyore_amd import _lib as L  # noqa: E402

PEAK = {torch.float32: (157.3, 146.4), torch.float64: (78.6, 46.4)}  # TFLOP/s: datasheet, measured 16x16x4 (tools/peak_probe.hip)


def f_step(dims, n_rows):
    prods = [dims[i] * dims[i + 1] for i in range(len(dims) - 1)]
    P = sum((dims[i] + 1) * dims[i + 1] for i in range(len(dims) - 1))
    return 2 * n_rows * (2 * sum(prods) + sum(prods[1:])) + 6 * P


xs, ys = synthetic.iris_shaped_arrays(seed=0)
for dims, acts, lik, dtype, C in (([4, 32, 32, 3], [1, 1, 0], 1, torch.float64, 4096),
                                  ([4, 16, 16, 3], [1, 1, 0], 1, torch.float32, 4096),
                                  ([4, 32, 32, 3], [2, 2, 0], 1, torch.float32, 4096),
                                  ([4, 64, 64, 3], [1, 1, 0], 1, torch.float32, 4096),
                                  ([4, 16, 16, 3], [1, 1, 0], 1, torch.float64, 4096),
                                  ([4, 32, 32, 1], [1, 1, 1], 0, torch.float32, 4096),
                                  # hidden widths off the tile grid (zero-padded to the next of 16 / 32 / 64)
                                  ([4, 20, 20, 3], [1, 1, 0], 1, torch.float32, 4096),
                                  ([4, 50, 40, 3], [1, 1, 0], 1, torch.float32, 4096),
                                  ([4, 24, 24, 3], [1, 1, 0], 1, torch.float64, 4096),
                                  # one hidden layer (the kernel's middle layer skipped)
                                  ([4, 16, 3], [1, 0], 1, torch.float32, 4096),
                                  ([4, 32, 3], [1, 0], 1, torch.float32, 4096),
                                  ([4, 64, 3], [1, 0], 1, torch.float32, 4096),
                                  ([4, 32, 3], [1, 0], 1, torch.float64, 4096)):
    y = ys if lik == 1 else ys[:, :1]
    pl = Plan(dims, [1] * (len(dims) - 1), acts, lik, dtype, dev)
    pl.set_data(torch.tensor(xs, dtype=dtype, device=dev), torch.tensor(y, dtype=dtype, device=dev))
    pl.set_prior(torch.zeros(pl.P), torch.full((pl.P,), float(np.sqrt(3.0))))
    th = 0.1 * pl.philox_normal(C, seed=0, it=0)
    t, g = pl.log_target_grad(th)
    Ls, step = 20, 0.02
    res = {}
    for name, flags, n, Cn in (("fused16", 0, 10, C), ("generic", L.EY_FORCE_GENERIC, 2, min(C, 1024))):
        thn, tn, gn = th[:Cn].clone(), t[:Cn].clone(), g[:Cn].clone()
        acc = []
        def fs():
            it[0] += 1
            acc.append(pl.hmc_step(thn, tn, gn, step, Ls, seed=3, it=it[0], flags=flags)["accepted"])
        dt = timed(fs, n)
        res[name] = Cn * Ls / dt
        a = torch.stack(acc[-n:]).float().mean().item()
        if name == "fused16":
            fl = f_step(dims, xs.shape[0]) * res[name] / 1e12
            pk = PEAK[dtype]
            print(f"{pl.kernel} HMC L={Ls} {Cn} chains MLP({'-'.join(map(str, dims))}) acts {acts} "
                  f"{'CE' if lik else 'BCE'} {str(dtype)[6:]}: {dt * 1e3:.3f} ms/draw -> {res[name]:.3e} leapfrog-steps/s x "
                  f"chains = {fl:.1f} TFLOP/s ({100 * fl / pk[0]:.1f}% of the {pk[0]} datasheet matrix peak, "
                  f"{100 * fl / pk[1]:.1f}% of the {pk[1]} measured), acceptance {a:.2f}")
        else:
            print(f"     generic kernel on {Cn} chains: {res[name]:.3e} leapfrog-steps/s x chains -> fused16 is "
                  f"{res['fused16'] / res[name]:.1f}x")
