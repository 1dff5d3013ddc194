"""Known-answer vectors of Philox4x32-10 (Random123 kat_vectors) for the oracle's restatement, and moments of the Gaussian map."""
import numpy as np

from oracle import rng as orng


def test_philox4x32_10_known_answers():
    # Random123 examples/kat_vectors: philox4x32 10 <ctr x4> <key x2> -> <out x4>
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = orng.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want


def test_normal_pairs_moments_and_independence():
    z = orng.normal_pairs(1234, np.arange(200000), 3)
    assert abs(z.mean()) < 7e-3 and abs(z.var() - 1.0) < 1e-2 and abs(np.mean(z[:, 0] * z[:, 1])) < 1e-2  # > 4 sigma
    assert abs(np.mean(z ** 4) - 3.0) < 0.06
    z2 = orng.normal_pairs(1234, np.arange(200000), 4)
    assert abs(np.mean(z[:, 0] * z2[:, 0])) < 1e-2  # another draw index is another stream
    assert np.array_equal(orng.normal_pairs(1234, np.arange(10), 3), z[:10])  # stateless
    assert not np.array_equal(orng.normal_pairs(1235, np.arange(10), 3), z[:10])
