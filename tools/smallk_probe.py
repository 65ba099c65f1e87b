#!/usr/bin/env python3
"""The layerwise path's 128-wide batched product on mid-size shapes (M rows x N x K per batch item, both operands k-contiguous as
in a forward layer), through ey_debug_bgemm: time against K (the number of 16-wide chunks) separates a workgroup's fixed cost
from its cost per chunk.   python tools/smallk_probe.py [batch]"""
import ctypes as ct, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L
dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
st = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in ((512, 100, 16), (512, 100, 32), (512, 100, 64), (512, 100, 96), (512, 100, 100), (512, 100, 112), (512, 100, 128),
                  (512, 100, 256), (512, 128, 128), (128, 128, 128), (2048, 100, 100)):
    A = torch.randn(batch, M, K, device=dev); B = torch.randn(batch, N, K, device=dev); C = torch.zeros(batch, M, N, device=dev)
    def run():
        L.check(L.lib().ey_debug_bgemm(L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, K, 1, 1, K, N, 1, M * K, N * K, M * N, None, 0, 0, batch, st), "bgemm")
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    wgs = ((M + 127) // 128) * ((N + 127) // 128) * batch
    print(f"M {M:4d} N {N:3d} K {K:3d} ({(K + 15) // 16:2d} chunks) batch {batch}: {ms * 1e3:8.1f} us  {2.0 * M * N * K * batch / ms / 1e9:6.1f} TFLOP/s  "
          f"{wgs} workgroups = {wgs / 768:.1f} rounds of 768: {ms * 1e3 / (wgs / 768):6.1f} us per round")
