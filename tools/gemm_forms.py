# error of the two forms of the 128-wide batched product against f64, and their speed (tools/gemm_probe.py shapes)
import ctypes as ct, sys, os
import torch
sys.path.insert(0, os.getcwd())
from eeyore_amd import _lib as L
dev = torch.device("cuda", 0)
torch.manual_seed(0)
st = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, kfast, batch) in ((1024, 128, 784, True, 512), (128, 768, 1024, False, 512), (256, 256, 256, True, 64), (300, 200, 100, False, 16)):
    if kfast:
        A = torch.randn(batch, M, K, device=dev); Bt = torch.randn(batch, N, K, device=dev)
        ref = torch.bmm(A[:8].double(), Bt[:8].double().transpose(1, 2)); mag = torch.bmm(A[:8].double().abs(), Bt[:8].double().abs().transpose(1, 2))
        sA, sB, bA, bB, a, b = (K, 1), (1, K), M * K, N * K, A, Bt
    else:
        At = torch.randn(batch, K, M, device=dev); B = torch.randn(batch, K, N, device=dev)
        ref = torch.bmm(At[:8].double().transpose(1, 2), B[:8].double()); mag = torch.bmm(At[:8].double().abs().transpose(1, 2), B[:8].double().abs())
        sA, sB, bA, bB, a, b = (1, M), (N, 1), K * M, K * N, At, B
    for name, v in (("bf16x3", 0), ("exact", 1024)):
        L.lib().ey_debug_set_variant(v)
        C = torch.zeros(batch, M, N, device=dev)
        def run():
            L.check(L.lib().ey_debug_bgemm(L.ptr(a), L.ptr(b), L.ptr(C), M, N, K, sA[0], sA[1], sB[0], sB[1], N, 1, bA, bB, M * N, None, 0, 0, batch, st), "bgemm")
        run(); run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        err = ((C[:8].double() - ref).abs() / (mag * 2.0 ** -24))
        print(f"M {M} N {N} K {K} kfast {kfast} batch {batch} {name:7s}: {ms:8.3f} ms = {2.0 * M * N * K * batch / ms / 1e9:7.1f} TFLOP/s   "
              f"error max {err.max().item():.3f} rms {err.pow(2).mean().sqrt().item():.4f} (units of 2^-24 sum|a||b|)")
    L.lib().ey_debug_set_variant(0)
