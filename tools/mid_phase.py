#!/usr/bin/env python3
"""Per-phase s_memtime sums of the fused mid-size kernel (diagnostic build):
    make -C eeyore_amd/csrc EXTRA=-DMID_TIMING=1 OBJDIR=/tmp/obj_midt OUT=../lib/libeeyore_amd_midt.so
    EEYORE_AMD_LIB=eeyore_amd/lib/libeeyore_amd_midt.so python tools/mid_phase.py 20,100,100,5 512 1024"""
import ctypes as ct
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402

dims = [int(v) for v in sys.argv[1].split(",")]
N, C = int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
x = rng.standard_normal((N, dims[0])).astype(np.float32)
y = np.eye(dims[-1], dtype=np.float32)[rng.integers(0, dims[-1], N)]
K = len(dims) - 1
pl = Plan(dims, [1] * K, [1] * (K - 1) + [0], 1, torch.float32, dev)
pl.set_data(torch.tensor(x, device=dev), torch.tensor(y, device=dev))
pl.set_prior(torch.zeros(pl.P), torch.ones(pl.P))
pl.set_variant(int(os.environ.get('MID_VARIANT', '8192')))
th = 0.1 * pl.philox_normal(C, seed=0, it=0)
lib = ct.CDLL(os.environ["EEYORE_AMD_LIB"])
buf = (ct.c_ulonglong * 32)()
for _ in range(3):
    pl.log_target_grad(th)
torch.cuda.synchronize()
lib.ey_debug_mid_phase_read(buf, 1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    pl.log_target_grad(th)
b.record()
torch.cuda.synchronize()
lib.ey_debug_mid_phase_read(buf, 1)
chains = buf[31]
names = ["staging", "data tile", "forward hidden", "partial logits", "loss", "output layer backward", "2nd hidden backward",
         "first layer dW", "combine + write-out"]
if os.environ.get("MID_VARIANT", "8192") == "0":  # k_mid32's stamps
    names = ["staging", "(waves without a tile)", "output layer + loss", "data tile + forward", "output layer backward", "hidden layers backward",
             "first layer dW", "-", "end: temperature, pointers", "end: barrier in front of a pass", "end: accumulators to LDS",
             "end: barrier behind them", "end: sums, prior gradient, stores", "end: log-likelihood"]
rounds = ((N + 31) // 32 + 1) // 2
print(f"{dims} N={N} C={C}: {a.elapsed_time(b) / 10 * 1e3:.1f} us per evaluation (whole, with the timing stamps); {chains} chains timed, "
      f"{rounds} rounds per chain; ticks per chain:")
tot = sum(buf[i] for i in range(len(names))) / max(1, chains)
for i, n in enumerate(names):
    v = buf[i] / max(1, chains)
    print(f"  {n:26s}{v:10.0f}  ({100 * v / tot:4.1f} %)" + (f"  = {v / rounds:8.0f} per round" if 1 <= i <= 7 else ""))
print(f"  total {tot:.0f}")
if any(buf[16 + w] for w in range(8)):
    print("  the tiles of a chain by wave: " + " ".join(f"{buf[16 + w] / max(1, chains):.0f}" for w in range(8)))
