#!/usr/bin/env python3
"""The batched f32 GEMM of the config-5 path on its own (ey_debug_bgemm): TFLOP/s of the two big products as the
contraction length grows (separates the main loop from prologue / epilogue cost), and of the narrow ones in GB/s.
usage: gemm_probe.py [batch]"""
import ctypes as ct
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L  # noqa: E402

dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def run(A, B, C, M, N, K, sA, sB, sC, bA, bB, bC, act=0, reps=5):
    st = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    def f():
        L.check(L.lib().ey_debug_bgemm(L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1],
                                       bA, bB, bC, None, 0, act, batch, st), "ey_debug_bgemm")
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for K in (784, 1568, 3136):
    # forward layer 0 shape: X [1024, K] shared, W [128, K] per chain (k contiguous), H [1024, 128]
    X = torch.randn(1024, K, device=dev)
    W = torch.randn(batch, 128, K, device=dev)
    H = torch.empty(batch, 1024, 128, device=dev)
    for act in (0, 1):
        dt = run(X, W, H, 1024, 128, K, (K, 1), (1, K), (128, 1), 0, 128 * K, 1024 * 128, act)
        print(f"k-fast  M=1024 N=128 K={K:5d} act={act} batch {batch}: {dt * 1e3:8.3f} ms  "
              f"{2 * 1024 * 128 * K * batch / dt / 1e12:6.1f} TFLOP/s")
for Kr in (1024, 2048, 4096):
    # weight-gradient shape: D^T [128 x Kr] per chain (m contiguous), X [Kr, 768] shared (n contiguous), dW [128, 768]
    D = torch.randn(batch, Kr, 128, device=dev)
    X = torch.randn(Kr, 768, device=dev)
    G = torch.empty(batch, 128, 768, device=dev)
    dt = run(D, X, G, 128, 768, Kr, (1, 128), (768, 1), (768, 1), Kr * 128, 0, 128 * 768)
    print(f"row-fast M=128 N=768 K={Kr:5d} batch {batch}: {dt * 1e3:8.3f} ms  "
          f"{2 * 128 * 768 * Kr * batch / dt / 1e12:6.1f} TFLOP/s")
