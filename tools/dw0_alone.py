#!/usr/bin/env python3
"""Config 5's first-layer weight-gradient product (128 x 768 x 1024 per chain, x shared) as a PLAIN product through
ey_debug_bgemm (the output stored, no prior gradient, no fused leapfrog update), 4096 chains: what the main loop costs
without the epilogue that config 5 gives it.  python tools/dw0_alone.py [chains]"""
import ctypes as ct, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L
dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M, N, K = 128, 768, 1024
A = torch.randn(batch, K, M, device=dev); B = torch.randn(1, K, 784, device=dev); C = torch.zeros(batch, M, N, device=dev)
st = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
bA = 0 if os.environ.get("DW0_SHARED_A") else K * M     # DW0_SHARED_A=1: every chain reads the same delta (cache-resident operands)
bC = 0 if os.environ.get("DW0_SHARED_C") else M * N     # DW0_SHARED_C=1: every chain writes the same output tile (no write traffic to speak of)
def run():
    L.check(L.lib().ey_debug_bgemm(L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, 1, M, 784, 1, N, 1, bA, 0, bC, None, 0, 0, batch, st), "bgemm")
run(); run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"plain product{' (shared A)' if not bA else ''}{' (shared C)' if not bC else ''}, {batch} chains: {ms:.3f} ms  {2.0 * M * N * K * batch / ms / 1e9:.1f} TFLOP/s")
