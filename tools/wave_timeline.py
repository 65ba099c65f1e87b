#!/usr/bin/env python3
"""When does each chain's wave start and end inside one launch of the fused kernel?  (diagnostic build, see
tools/phase_timing.py; s_memrealtime = 100 MHz)   EEYORE_AMD_LIB=tools/abl/lib_phase.so python tools/wave_timeline.py"""
import ctypes as ct
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L  # noqa: E402
from eeyore_amd.datasets import synthetic  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402

dev = torch.device("cuda", 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
plan = Plan([4, 32, 32, 3], [1, 1, 1], [1, 1, 0], 1, torch.float32, dev)
plan.set_data(torch.tensor(xs, dtype=torch.float32, device=dev), torch.tensor(ys, dtype=torch.float32, device=dev))
plan.set_prior(torch.zeros(plan.P), torch.full((plan.P,), float(np.sqrt(3.0))))
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
plan.set_variant(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
theta = 0.1 * plan.philox_normal(C, seed=0, it=0)
target, grad = plan.log_target_grad(theta)
for it in range(5):
    plan.hmc_step(theta, target, grad, 0.024, 20, seed=1, it=1 + it)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
plan.hmc_step(theta, target, grad, 0.024, 20, seed=1, it=9)
b.record()
torch.cuda.synchronize()
buf = (ct.c_ulonglong * (3 * C))()
L.lib().ey_debug_wave_times(buf, C)
t = np.array(buf, dtype=np.float64).reshape(C, 3) / 100.0  # microseconds
t0 = t[:, 0].min()
entry, start, end = t[:, 0] - t0, t[:, 1] - t0, t[:, 2] - t0
print(f"launch {a.elapsed_time(b) * 1e3:.1f} us by events; first block entry -> last wave end {end.max():.1f} us")
first = start < np.median(start)
for name, m in (("first-round waves", first), ("second-round waves", ~first)):
    print(f"{name}: {m.sum()}  block entry {entry[m].min():.1f}..{entry[m].max():.1f} us, wave start "
          f"{start[m].min():.1f}..{start[m].max():.1f}, end {end[m].min():.1f}..{end[m].max():.1f}, lifetime "
          f"{(end - start)[m].mean():.1f} +- {(end - start)[m].std():.1f} us (min {(end - start)[m].min():.1f}, max {(end - start)[m].max():.1f})")
print("percentiles of wave end (us):", np.percentile(end, [1, 10, 25, 50, 75, 90, 99, 100]).round(1))
print("percentiles of second-round block entry (us):", np.percentile(entry[~first], [0, 1, 10, 50, 90, 99, 100]).round(1))
dbg = (ct.c_int * 64)()
L.lib().ey_debug_ints(dbg)
print("block 0: simd ids", list(dbg[0:8]), "partners", list(dbg[8:16]), "mates", list(dbg[16:24]))
print("HW_ID", [hex(v & 0xffffffff) for v in dbg[24:32]])
if (int(sys.argv[2]) if len(sys.argv) > 2 else 0) & 1 == 0:
    grid = min(256, C)
    wv = (np.arange(C) // grid) % 8
    rnd = np.arange(C) // (grid * 8)
    for r in range(int(rnd.max()) + 1):
        print(f"round {r}: mean lifetime by wave index (us):",
              [round(float((end - start)[(wv == w) & (rnd == r)].mean()), 1) for w in range(8)])
