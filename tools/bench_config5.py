#!/usr/bin/env python3
"""BASELINE config 5: parallel-tempering HMC (L = 20) on MLP(784-128-10), MNIST-shaped synthetic data (N = 1024 rows,
~19 % non-zero pixels, 10 balanced classes).  One ladder position per GPU, R chains (replicas) per GPU, ladder
t_i = (i/K)^4, even/odd neighbour exchange of temperature labels every 10 iterations (one RCCL all-gather of [R] log-targets
and labels; no state crosses xGMI).

    python tools/bench_config5.py [chains] [iterations]                              # one GPU: one temperature
    python -m torch.distributed.run --nproc-per-node 8 tools/bench_config5.py 4096 20  # the full configuration

Reports leapfrog-steps/s x chains and TFLOP/s on the algorithmic flops of SURVEY.md 8(d) (4.097e8 per step per chain)."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.distributed import TemperingExchange, init_from_env  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402

rank, world, local = init_from_env()
local = local % max(1, torch.cuda.device_count())
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if os.environ.get("EY_NO_DMA"):  # A/B: the register-staged GEMM instead of the LDS-DMA one
    from eeyore_amd import _lib as L
    L.lib().ey_debug_set_variant(32)
if os.environ.get("EY_VARIANT"):  # A/B: any ey_debug_set_variant bits (4096 = one chain per workgroup in the shared-operand product)
    from eeyore_amd import _lib as L
    L.lib().ey_debug_set_variant(int(os.environ["EY_VARIANT"]))
N, L, between = 1024, 20, 10
# the step: EY_CFG5_STEP (default 0.024: acceptance 0.74 after burn-in, tools/step_sweep_cfg5.sh -> profiles/r04_cfg5_step_sweep.txt; 0.016: 0.89, 0.032: 0.00)
eps = float(os.environ.get("EY_CFG5_STEP", "0.024"))
burnin = int(os.environ.get("EY_CFG5_BURNIN", "20"))   # untimed iterations before the clock (acceptance is reported over the timed ones)
DT = torch.float64 if os.environ.get("EY_F64") else torch.float32   # EY_F64=1: the f64 layerwise path (parity dtype)
PEAK, PEAK_NAME = (78.6e12, "f64") if DT == torch.float64 else (157.3e12, "f32")

rng = np.random.default_rng(0)
x = (rng.random((N, 784)) * (rng.random((N, 784)) < 0.19)).astype(np.float32)
y = np.eye(10, dtype=np.float32)[np.arange(N) % 10]
pl = Plan([784, 128, 10], [1, 1], [1, 0], 1, DT, dev)
pl.set_data(torch.tensor(x, device=dev, dtype=DT), torch.tensor(y, device=dev, dtype=DT))
pl.set_prior(torch.zeros(pl.P), torch.ones(pl.P))

ladder = [(i / world) ** 4 for i in range(1, world + 1)]
pt = TemperingExchange(ladder, C, rank, world, dev, seed=11)
th = 0.05 * pl.philox_normal(C, seed=0, it=0)  # every rank starts its replicas from the same states
temps = pt.temperature_vector(DT)
t, g = pl.log_target_grad(th, temp=temps)
out = pl.hmc_step(th, t, g, eps, L, temp=temps, seed=1 + rank, it=1)
for b_ in range(burnin):
    out = pl.hmc_step(th, t, g, eps, L, temp=temps, seed=1 + rank, it=1000000 + b_)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
accs, swaps = [], 0
for it in range(iters):
    out = pl.hmc_step(th, t, g, eps, L, temp=temps, seed=1 + rank, it=2 + it)
    accs.append(out["accepted"].float().mean().item())
    if world > 1 and (it + 1) % between == 0:
        old = temps
        swaps += pt.exchange(t / old)          # untempered log-target ell = T / t
        temps = pt.temperature_vector(DT)
        ratio = temps / old                    # a relabelled chain keeps its state: rescale the cached tempered values
        t *= ratio
        g *= ratio[:, None]
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = (time.perf_counter() - t0) / iters
f_step = 2 * N * (2 * (784 * 128 + 128 * 10) + 128 * 10) + 6 * pl.P
if rank == 0:
    tot = C * world
    print(f"kernel {pl.kernel}: {world} temperature(s) x {C} chains, {dt * 1e3:.1f} ms per HMC iteration (L={L}) -> "
          f"{tot * L / dt:.3e} leapfrog-steps/s x chains, {f_step * C * L / dt / 1e12:.1f} TFLOP/s per GPU "
          f"({100 * f_step * C * L / dt / PEAK:.1f}% of {PEAK_NAME} MFMA peak), step {eps}, acceptance {np.mean(accs):.2f} "
          f"(after {burnin} burn-in iterations), "
          f"label exchanges accepted {swaps}")
if world > 1:
    dist.destroy_process_group()
