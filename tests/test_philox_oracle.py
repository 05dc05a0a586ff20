"""CPU: the NumPy restatement of Philox4x32-10 (oracle/philox.py) against the known-answer vectors of Random123
(``kat_vectors``, lines ``philox4x32 10``), and sanity of the transforms."""
import numpy as np

from oracle import philox as ph

# Random123 examples/kat_vectors: philox4x32 10  <counter x4> <key x2> -> <output x4>
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_known_answer_vectors():
    for ctr, key, want in KAT:
        got = ph.philox4x32_10(*ctr, *key)
        assert tuple(int(g) for g in got) == want


def test_vectorised_equals_scalar_and_counters_are_independent():
    m = np.arange(1000)
    r = ph.philox_at(12345, ph.STREAM_DW, 7, m)
    for i in (0, 1, 999):
        s = ph.philox_at(12345, ph.STREAM_DW, 7, i)
        assert all(int(a[i]) == int(b) for a, b in zip(r, s))
    r2 = ph.philox_at(12345, ph.STREAM_DW, 8, m)
    assert not np.array_equal(r[0], r2[0])
    # 64-bit sample indices reach counter word 1
    hi = ph.philox_at(1, 1, 0, np.array([5, 5 + (1 << 32)], dtype=np.uint64))
    assert int(hi[0][0]) != int(hi[0][1])


def test_transforms_ranges_and_moments():
    r = ph.philox_at(3, ph.STREAM_USER, np.arange(64)[:, None], np.arange(4096)[None, :])
    u = ph.u01(r[0])
    assert u.min() > 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 3e-3 and abs(u.var() - 1 / 12) < 2e-3
    n0, n1 = ph.box_muller(r[0], r[1])
    z = np.concatenate([n0.ravel(), n1.ravel()])
    assert np.isfinite(z).all() and np.abs(z).max() < 5.8            # sqrt(-2 ln 2^-24) = 5.77
    assert abs(z.mean()) < 5e-3 and abs(z.var() - 1.0) < 1e-2
    assert abs(np.mean(n0 * n1)) < 5e-3


def test_sampler_layouts():
    DWs, masses, obs_Qs = ph.drone_sample(7, 33, 5, 2.5)
    assert DWs.shape == (33, 5, 6) and np.all(DWs[:, :, :3] == 0)
    assert masses.min() >= 29 and masses.max() <= 35
    r = 1 / np.sqrt(obs_Qs[:, 0, 0, 0])
    assert r.min() >= 0.275 and r.max() <= 0.325
    a, th, tau = ph.hopper_sample(1, 17)
    assert a.shape == (17, 30) and th.max() < np.pi and tau.max() < 2 * np.pi
