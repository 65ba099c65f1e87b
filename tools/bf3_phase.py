#!/usr/bin/env python3
"""Where a workgroup of the bf16x3 128 x 128 product spends its time: s_memtime sums per phase of the chunk loop, wave 0 of every
16th workgroup, from a library built with -DEY_BF3_TIMING=1 (EEYORE_AMD_LIB).  The product is config 5's first-layer weight
gradient as a plain product (tools/dw0_alone.py); DW0_SHARED_A / DW0_SHARED_C as there.  python tools/bf3_phase.py [chains]"""
import ctypes as ct, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L
dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
st = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
fwd = os.environ.get("BF3_FORWARD")  # BF3_FORWARD=M,N,K: a forward product act(A[b] B[b]^T + bias[b]), both operands k-contiguous, per chain
if fwd:
    M, N, K = (int(v) for v in fwd.split(","))
    A = torch.randn(batch, M, K, device=dev); B = torch.randn(batch, N, K, device=dev); C = torch.zeros(batch, M, N, device=dev)
    bias = torch.randn(batch, N, device=dev)
    bA = bC = 1
    def run():
        if os.environ.get("BF3_PLAIN"):  # no bias, no activation
            L.check(L.lib().ey_debug_bgemm(L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, K, 1, 1, K, N, 1, M * K, N * K, M * N, None, 0, 0, batch, st), "bgemm")
        else:
            L.check(L.lib().ey_debug_bgemm(L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, K, 1, 1, K, N, 1, M * K, N * K, M * N, L.ptr(bias), N, 1, batch, st), "bgemm")
else:
    M, N, K = 128, 768, 1024
    A = torch.randn(batch, K, M, device=dev); B = torch.randn(1, K, 784, device=dev); C = torch.zeros(batch, M, N, device=dev)
    bA = 0 if os.environ.get("DW0_SHARED_A") else K * M
    bC = 0 if os.environ.get("DW0_SHARED_C") else M * N
    def run():
        L.check(L.lib().ey_debug_bgemm(L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, 1, M, 784, 1, N, 1, bA, 0, bC, None, 0, 0, batch, st), "bgemm")
run(); torch.cuda.synchronize()
buf = (ct.c_ulonglong * 16)()
assert L.lib().ey_debug_bf3_phase_read(buf, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
assert L.lib().ey_debug_bf3_phase_read(buf, 1) == 0
v = list(buf)
n, chunks = v[10], v[11] / max(1, v[10])
print(f"{'forward ' + fwd + ', ' if fwd else ''}{batch} chains{' shared A' if not bA else ''}{' shared C' if not bC else ''}: {e0.elapsed_time(e1):.3f} ms with the stamps; {n} workgroups timed, {chunks:.0f} chunks each; "
      f"s_memtime ticks (shader clock cycles) per chunk:")
names = ["loop control", "fetches issued", "fragments read (ds_read + wait)", "24 products issued", "split + staging stores (vmcnt wait inside)", "barrier"]
tot = sum(v[:6])
for i, nm in enumerate(names):
    print(f"  {nm:44s} {v[i] / n / chunks:8.1f}  ({100.0 * v[i] / tot:5.1f} %)")
print(f"  chunk loop per workgroup {v[8] / n:10.0f} ticks, row sums + epilogue {v[9] / n:8.0f}")
