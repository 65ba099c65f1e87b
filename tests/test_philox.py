"""The counter-based random streams: Philox4x32-10 against Random123's published known-answer vectors (numpy twin
and the library's host entry point), and the device streams against the twin -- integer words and uniforms bit for
bit, normals to within the libm differences of log / sqrt / sinpi / cospi."""
import ctypes as ct

import numpy as np
import pytest

from oracle import philox_oracle as po


def _host_block(ctr, key):
    from eeyore_amd import _lib as L
    c = (ct.c_uint32 * 4)(*ctr)
    k = (ct.c_uint32 * 2)(*key)
    o = (ct.c_uint32 * 4)()
    L.check(L.lib().ey_philox_block(c, k, o), "ey_philox_block")
    return tuple(int(v) for v in o)


def test_numpy_twin_reproduces_random123_kat():
    for ctr, key, exp in po.RANDOM123_KAT:
        got = tuple(int(v) for v in po.philox4x32_10(*ctr, *key))
        assert got == exp


def test_library_host_philox_reproduces_random123_kat_and_twin():
    for ctr, key, exp in po.RANDOM123_KAT:
        assert _host_block(ctr, key) == exp
    rng = np.random.default_rng(5)
    for _ in range(200):
        ctr = [int(v) for v in rng.integers(0, 2 ** 32, 4, dtype=np.uint64)]
        key = [int(v) for v in rng.integers(0, 2 ** 32, 2, dtype=np.uint64)]
        assert _host_block(ctr, key) == tuple(int(v) for v in po.philox4x32_10(*ctr, *key))


def test_twin_streams_are_layout_independent():
    """Element i of a chain's stream is component i & 3 of block i >> 2 whatever P is asked for."""
    a = po.normal(3, 1315, seed=9, it=4, chain_offset=2 ** 33 + 5)
    b = po.normal(3, 1313, seed=9, it=4, chain_offset=2 ** 33 + 5)
    assert np.array_equal(a[:, :1313], b)
    assert not np.array_equal(po.normal(1, 8, 9, 4), po.normal(1, 8, 9, 5))
    assert not np.array_equal(po.normal(1, 8, 9, 4), po.normal(1, 8, 10, 4))
    u = po.uniform(5, 9, 4, dtype=np.float64)
    assert ((u >= 0) & (u < 1)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_device_streams_match_the_twin(dtype):
    import torch
    from eeyore_amd import _lib as L
    dev = torch.device("cuda", 0)
    tdt, ndt = (torch.float32, np.float32) if dtype == "f32" else (torch.float64, np.float64)
    code = L.EY_F32 if dtype == "f32" else L.EY_F64
    stream = ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for C, P, seed, it, off in ((7, 1315, 2024, 1, 0), (3, 9, 0xDEADBEEFCAFE, 2 ** 32 + 17, 2 ** 33 + 4096), (64, 20, 1, 0, 5)):
        out = torch.empty(C, P, dtype=tdt, device=dev)
        L.check(L.lib().ey_philox_normal(L.ptr(out), C, P, seed, it, off, code, stream), "ey_philox_normal")
        uo = torch.empty(C, dtype=tdt, device=dev)
        L.check(L.lib().ey_philox_uniform(L.ptr(uo), C, seed, it, off, code, stream), "ey_philox_uniform")
        torch.cuda.synchronize()
        # uniforms: an integer scaled by a power of two -- bit for bit
        assert np.array_equal(uo.cpu().numpy(), po.uniform(C, seed, it, off, ndt))
        want = po.normal(C, P, seed, it, off, ndt)
        got = out.cpu().numpy()
        # normals: r = sqrt(-2 log u1) and sin/cos of an exact angle; a few ulp of the f32 / f64 libm
        tol = 4e-6 if dtype == "f32" else 1e-13
        np.testing.assert_allclose(got, want, rtol=tol, atol=tol)
    # the moments of a long stream (a wrong Box-Muller pairing would show here)
    big = torch.empty(256, 4096, dtype=tdt, device=dev)
    L.check(L.lib().ey_philox_normal(L.ptr(big), 256, 4096, 11, 3, 0, code, stream), "ey_philox_normal")
    b = big.double()
    assert abs(b.mean().item()) < 5e-3 and abs(b.var().item() - 1.0) < 5e-3
    assert abs((b ** 4).mean().item() - 3.0) < 5e-2
