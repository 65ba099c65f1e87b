"""CPU twin of the library's counter-based random streams.  TEST INFRASTRUCTURE ONLY: imported by tests/, never by
eeyore_amd/.

The reference draws from torch's global Mersenne-Twister stream (eeyore/samplers/hmc.py:134,148, mala.py:53,66,
metropolis_hastings.py:45,56), which a chain-batched kernel cannot reproduce ("RNG"); the C ABI
therefore takes the variates as inputs (parity mode) or draws them from Philox4x32-10 (Salmon, Moraes, Dror, Shaw:
"Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 v1.14 `philox.h`), keyed as
eeyore_amd/csrc/ey_common.h lays out:

    key     = (seed_lo, seed_hi)
    counter = (block, chain_lo, iter_lo, (iter_hi << 8) | stream | (chain_hi << 20))
    stream 0: N(0,1) -- f32: ONE call per block of four elements, two Box-Muller pairs on 24-bit uniforms
                        f64: two calls per block (counters 2b, 2b+1), one pair each on 53-bit uniforms
    stream 1: U[0,1) accept variate (block 0, word 0 [and word 1 for f64])

`philox4x32_10` is pinned by Random123's published known-answer vectors (tests/test_philox.py); the integer words and
the uniforms must match the device bit for bit, the normals to within the libm differences of log / sqrt / sinpi / cospi.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
STREAM_NORMAL, STREAM_UNIFORM = 0, 1
_MASK = np.uint64(0xFFFFFFFF)

# Random123 v1.14, examples/kat_vectors: "philox4x32 10  <ctr x4> <key x2>  <expected x4>"
RANDOM123_KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Ten rounds on arrays of uint32 counters / keys (broadcast together); returns four uint32 arrays."""
    c0, c1, c2, c3, k0, k1 = np.broadcast_arrays(*[np.asarray(a, dtype=np.uint32) for a in (c0, c1, c2, c3, k0, k1)])
    c0, c1, c2, c3, k0, k1 = (a.copy() for a in (c0, c1, c2, c3, k0, k1))
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = k0 + W0
            k1 = k1 + W1
    return c0, c1, c2, c3


def _key_counter(seed, chain, it, stream):
    seed, chain, it = np.uint64(seed), np.asarray(chain, dtype=np.uint64), np.uint64(it)
    k0, k1 = np.uint32(seed & _MASK), np.uint32(seed >> np.uint64(32))
    c1 = (chain & _MASK).astype(np.uint32)
    c2 = np.uint32(it & _MASK)
    c3 = ((np.uint32(it >> np.uint64(32)) << np.uint32(8)) | np.uint32(stream & 0xff)) | \
        ((chain >> np.uint64(32)).astype(np.uint32) << np.uint32(20))
    return k0, k1, c1, c2, c3


def uniform(C, seed, it, chain_offset=0, dtype=np.float32):
    """out[c] of ey_philox_uniform: the accept variate of chain chain_offset + c at iteration `it`."""
    chain = np.arange(C, dtype=np.uint64) + np.uint64(chain_offset)
    k0, k1, c1, c2, c3 = _key_counter(seed, chain, it, STREAM_UNIFORM)
    o0, o1, _, _ = philox4x32_10(np.uint32(0), c1, c2, c3, k0, k1)
    if dtype == np.float32:
        return (o0 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return ((o0.astype(np.uint64) << np.uint64(21)) | (o1 >> np.uint32(11)).astype(np.uint64)).astype(np.float64) * 2.0 ** -53


def normal(C, P, seed, it, chain_offset=0, dtype=np.float32):
    """out[c, i] of ey_philox_normal (what p0 / z = NULL draws in the step kernels)."""
    chain = (np.arange(C, dtype=np.uint64) + np.uint64(chain_offset))[:, None]
    nb = (P + 3) // 4
    block = np.arange(nb, dtype=np.uint32)[None, :]
    k0, k1, c1, c2, c3 = _key_counter(seed, chain, it, STREAM_NORMAL)
    out = np.empty((C, nb, 4), dtype=dtype)
    if dtype == np.float32:
        o = philox4x32_10(block, c1, c2, c3, k0, k1)
        for pair in range(2):
            u1 = ((o[2 * pair] >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
            u2 = (o[2 * pair + 1] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
            rad = np.sqrt(np.float32(-2.0) * np.log(u1))
            ang = (np.float32(2.0) * u2).astype(np.float64) * np.pi  # cospif / sinpif: the argument in half-turns
            out[:, :, 2 * pair] = rad * np.cos(ang).astype(np.float32)
            out[:, :, 2 * pair + 1] = rad * np.sin(ang).astype(np.float32)
    else:
        for pair in range(2):
            o = philox4x32_10(np.uint32(2) * block + np.uint32(pair), c1, c2, c3, k0, k1)
            u1 = (((o[0].astype(np.uint64) << np.uint64(21)) | (o[2] >> np.uint32(11)).astype(np.uint64)).astype(np.float64)
                  + 0.5) * 2.0 ** -53
            u2 = ((o[1].astype(np.uint64) << np.uint64(21)) | (o[3] >> np.uint32(11)).astype(np.uint64)).astype(np.float64) \
                * 2.0 ** -53
            rad = np.sqrt(-2.0 * np.log(u1))
            out[:, :, 2 * pair] = rad * np.cos(2.0 * np.pi * u2)
            out[:, :, 2 * pair + 1] = rad * np.sin(2.0 * np.pi * u2)
    return out.reshape(C, nb * 4)[:, :P]
