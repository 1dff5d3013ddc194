"""ORACLE (test infrastructure only): the engine's counter-based Gaussian noise, restated in numpy.

Not a restatement of the reference (jax's threefry stream cannot be reproduced without jax): it mirrors
gddim_amd/csrc/kernels/rng.h -- Philox4x32-10 with counter (element lo, element hi, draw, 0) and key = seed, words 0/1 through
Box-Muller -- so that the stochastic samplers can be compared trajectory by trajectory with the same noise.  The Philox rounds
follow the published algorithm (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) and are checked
against its known-answer vectors in tests/test_oracle_rng.py."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays; returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def normal_pairs(seed, elems, draw, dtype=np.float64):
    """(len(elems), 2) standard normals of elements `elems` (uint64 indices) for draw index `draw`."""
    elems = np.asarray(elems, dtype=np.uint64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    r0, r1, _, _ = philox4x32_10((elems & _MASK).astype(np.uint32), (elems >> np.uint64(32)).astype(np.uint32),
                                 np.full(elems.shape, draw, dtype=np.uint32), np.zeros(elems.shape, dtype=np.uint32),
                                 seed & 0xFFFFFFFF, seed >> 32)
    u1 = ((r0 >> np.uint32(8)).astype(dtype) + 0.5) * 2.0 ** -24
    u2 = ((r1 >> np.uint32(8)).astype(dtype) + 0.5) * 2.0 ** -24
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * np.pi * u2
    return np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=-1)


def noise_like(seed, shape, draw, dtype=np.float64):
    """Noise tensor of CLD state shape (B, H, W, C, 2): element index = flat index over (B, H, W, C)."""
    n = int(np.prod(shape[:-1]))
    return normal_pairs(seed, np.arange(n, dtype=np.uint64), draw, dtype).reshape(tuple(shape))
