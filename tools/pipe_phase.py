#!/usr/bin/env python3
"""Per-phase s_memtime sums of the pipelined tile loop (eval_pipe; diagnostic build with -DEY_PHASE_TIMING=1):
    make -C eeyore_amd/csrc EXTRA=-DEY_PHASE_TIMING=1 OBJDIR=../lib/obj_phase OUT=../lib/libeeyore_amd_phase.so
    EEYORE_AMD_LIB=eeyore_amd/lib/libeeyore_amd_phase.so python tools/pipe_phase.py         (on the GPU box)"""
import ctypes as ct
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L  # noqa: E402
from eeyore_amd.datasets import synthetic  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402

NAMES = ["ph0 F0, act, split H0 (t+1)", "ph1 F1(t+1) || delta1(t), split", "ph2 dH0(t) || H1(t+1) = act", "ph3 tr loads; logits (4x4x1)",
         "ph4 dW1(t) || delta0(t), softmax(t+1)", "ph5 dW0(t) (4x4x1)", "ph6 dW2, dH1 (t+1) (4x4x1)", "-", "-", "-",
         "after the tile loop (per evaluation)"]
dev = torch.device("cuda", 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
plan = Plan([4, 32, 32, 3], [1, 1, 1], [1, 1, 0], 1, torch.float32, dev)
plan.set_data(torch.tensor(xs, dtype=torch.float32, device=dev), torch.tensor(ys, dtype=torch.float32, device=dev))
plan.set_prior(torch.zeros(plan.P), torch.full((plan.P,), float(np.sqrt(3.0))))
plan.set_variant(8)
lib = L.lib()
buf = (ct.c_ulonglong * 32)()
C = 4096
theta = 0.1 * plan.philox_normal(C, seed=0, it=0)
target, grad = plan.log_target_grad(theta)
for it in range(3):
    plan.hmc_step(theta, target, grad, 0.011, 20, seed=1, it=1 + it)
torch.cuda.synchronize()
lib.ey_debug_phase_read_p(buf, 1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for it in range(5):
    plan.hmc_step(theta, target, grad, 0.011, 20, seed=1, it=10 + it)
b.record()
torch.cuda.synchronize()
lib.ey_debug_phase_read_p(buf, 1)
evals = buf[15]
tot = np.array([buf[i] for i in range(11)], dtype=np.float64) / evals
print(f"{a.elapsed_time(b) / 5:.4f} ms per draw, {evals} evaluations timed; s_memtime ticks per EVALUATION (5 tiles: 1 prologue body, "
      f"4 steady bodies, 1 epilogue body): total {tot.sum():.0f}; per wave lifetime {buf[11] / max(1, buf[14]):.0f} over {buf[14]} chains")
for i, n in enumerate(NAMES):
    if n != "-":
        print(f"{n:42s}{tot[i]:10.0f}")
outer = np.array([buf[16 + i] for i in range(6)], dtype=np.float64) / max(1, buf[14])
print("per chain (ticks): " + "  ".join(f"{k} {v:.0f}" for k, v in zip(["prologue", "theta axpy x20", "images x20", "eval x20", "p axpy x20", "epilogue"], outer)))
