#!/usr/bin/env python3
"""The two forms of the layerwise path's 128-wide batched product (EY_OPT_F32_PRODUCTS: bf16x3 / exact f32 MFMA) through
ey_debug_bgemm: time, algorithmic TFLOP/s and error against f64 (units of 2^-24 sum_k |a||b|), for config 5's two big
products and two small ones, with none / one / both operands shared by every batch item (batch stride 0: in config 5 the
data matrix X is shared; with both shared everything is cache-resident and only the products remain).
Run on the GPU box:  python tools/gemm_forms.py"""
import ctypes as ct
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
st = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = ((1024, 128, 784, True, 512), (128, 768, 1024, False, 512), (256, 256, 256, True, 64), (300, 200, 100, False, 16))
for (M, N, K, kfast, batch), share in itertools.product(SHAPES, ("", "A", "B", "AB")):
    if kfast:
        A = torch.randn(batch, M, K, device=dev); Bm = torch.randn(batch, N, K, device=dev)
        sA, sB, bA, bB = (K, 1), (1, K), M * K, N * K
        op = lambda a, b: torch.bmm(a, b.transpose(1, 2))
    else:
        A = torch.randn(batch, K, M, device=dev); Bm = torch.randn(batch, K, N, device=dev)
        sA, sB, bA, bB = (1, M), (N, 1), K * M, K * N
        op = lambda a, b: torch.bmm(a.transpose(1, 2), b)
    a8 = A[:1].expand(8, -1, -1) if "A" in share else A[:8]
    b8 = Bm[:1].expand(8, -1, -1) if "B" in share else Bm[:8]
    ref, mag = op(a8.double(), b8.double()), op(a8.double().abs(), b8.double().abs())
    if "A" in share: bA = 0
    if "B" in share: bB = 0
    line = f"M {M:4d} N {N:4d} K {K:4d} {'k' if kfast else 'row'}-contiguous batch {batch:3d} shared '{share:2s}':"
    for name, v in (("bf16x3", 0), ("exact", 1024)):
        L.lib().ey_debug_set_variant(v)
        C = torch.zeros(batch, M, N, device=dev)

        def run():
            L.check(L.lib().ey_debug_bgemm(L.ptr(A), L.ptr(Bm), L.ptr(C), M, N, K, sA[0], sA[1], sB[0], sB[1], N, 1, bA, bB,
                                           M * N, None, 0, 0, batch, st), "bgemm")
        run(); run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        err = (C[:8].double() - ref).abs() / (mag * 2.0 ** -24)
        line += f"  {name} {ms:7.3f} ms {2.0 * M * N * K * batch / ms / 1e9:6.1f} TF/s err max {err.max().item():.2f} rms {err.pow(2).mean().sqrt().item():.3f}"
    L.lib().ey_debug_set_variant(0)
    print(line)
