#!/usr/bin/env python3
"""Per-phase s_memtime sums of the fused kernel's tile loop (diagnostic build with -DEY_PHASE_TIMING=1).

    hipcc ... -DEY_PHASE_TIMING=1 -shared -o tools/abl/lib_phase.so eeyore_amd/csrc/*.hip       (built here)
    EEYORE_AMD_LIB=tools/abl/lib_phase.so python tools/phase_timing.py [chains ...]             (on the GPU box)

4096 chains = two waves per SIMD (the production occupancy); 1024 chains = one wave per SIMD, i.e. what one wave costs
when nothing shares its ALU: the difference between the two columns is what the partner wave hides."""
import ctypes as ct
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L  # noqa: E402
from eeyore_amd.datasets import synthetic  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402

NAMES = ["F0 + sigmoid + store H0^T", "F1 + sigmoid + store H1^T", "logits (4x4x1) + half-sum", "softmax/CE + delta2 store",
         "dW2 (reads + 4x4x1)", "dH1 (4x4x1) + delta1", "delta1 transpose store", "dW1 (reads + 16 MFMA)",
         "dH0 (16 MFMA) + delta0", "dW0 (4x4x1) + fence", "after the tile loop (x1/5 per tile)"]
dev = torch.device("cuda", 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
plan = Plan([4, 32, 32, 3], [1, 1, 1], [1, 1, 0], 1, torch.float32, dev)
plan.set_data(torch.tensor(xs, dtype=torch.float32, device=dev), torch.tensor(ys, dtype=torch.float32, device=dev))
plan.set_prior(torch.zeros(plan.P), torch.full((plan.P,), float(np.sqrt(3.0))))
lib = L.lib()
buf = (ct.c_ulonglong * 32)()
cols, outer = {}, {}
for C in [int(a) for a in sys.argv[1:]] or [4096, 1024]:
    theta = 0.1 * plan.philox_normal(C, seed=0, it=0)
    target, grad = plan.log_target_grad(theta)
    for it in range(3):
        plan.hmc_step(theta, target, grad, 0.024, 20, seed=1, it=1 + it)
    torch.cuda.synchronize()
    lib.ey_debug_phase_read(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for it in range(5):
        plan.hmc_step(theta, target, grad, 0.024, 20, seed=1, it=10 + it)
    b.record()
    torch.cuda.synchronize()
    lib.ey_debug_phase_read(buf, 1)
    evals = buf[15]
    per_tile = np.array([buf[i] for i in range(11)], dtype=np.float64) / (evals * 5)
    cols[C] = (per_tile, a.elapsed_time(b) / 5, evals, buf[11] / max(1, buf[14]))
    outer[C] = np.array([buf[16 + i] for i in range(6)], dtype=np.float64) / max(1, buf[14])
for C, (pt, ms, evals, kt) in cols.items():
    print(f"chains {C}: {ms:.4f} ms per draw, {evals} evaluations timed, s_memtime ticks per tile {pt.sum():.0f}; "
          f"{kt:.0f} ticks per wave lifetime: {100 * pt[:10].sum() * 100 / kt:.1f}% in the tile loop, "
          f"{100 * pt[10] * 100 / kt:.1f}% in the evaluation's epilogue, the rest between evaluations")
print(f"{'phase':34s}" + "".join(f"{C:>12d}" for C in cols))
for i, n in enumerate(NAMES):
    print(f"{n:34s}" + "".join(f"{cols[C][0][i]:12.0f}" for C in cols))
print("per chain (ticks): " + "  ".join(["prologue", "theta axpy x20", "images x20", "eval x20", "p axpy x20", "epilogue"]))
for C in cols:
    print(f"{C:6d}: " + "  ".join(f"{v:10.0f}" for v in outer[C]))
