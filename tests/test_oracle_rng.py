"""The oracle's random sources: numpy-legacy primitives vs the installed numpy (third-party dependency of
the reference, stream frozen by NEP 19) and Philox4x32-10 vs the Random123 known-answer vectors."""
import numpy as np

from oracle import oracle as orc


def test_mt19937_matches_numpy_randomstate():
    for seed in (0, 1, 5, 12345, 2**32 - 1):
        rs = np.random.RandomState(seed)
        mt = orc.MT(seed)
        want = rs.randint(0, 2**32, size=2000, dtype=np.uint32)
        got = np.array([mt.u32() for _ in range(2000)], np.uint32)
        np.testing.assert_array_equal(got, want)


def test_random_sample_and_interval_and_permutation():
    rs = np.random.RandomState(7)
    mt = orc.MT(7)
    for _ in range(50):
        assert mt.double() == rs.random_sample()
    for hi in (2, 4, 10, 41, 6400, 5321):
        want = rs.randint(0, hi, size=64)
        got = [mt.interval(hi - 1) for _ in range(64)]
        np.testing.assert_array_equal(got, want)
    for n in (1, 2, 4, 6, 100, 6400, 6089):
        np.testing.assert_array_equal(mt.permutation(n), rs.permutation(n))
    # choice(n, k, replace=False) == permutation(n)[:k], including k == 0 (still shuffles)
    for n, k in ((6400, 320), (5000, 2), (4, 1), (5000, 0)):
        want = rs.choice(n, size=k, replace=False)
        got = mt.permutation(n)[:k]
        np.testing.assert_array_equal(got, want)
    assert mt.u32() == int(rs.randint(0, 2**32, dtype=np.uint32))


def test_randint_one_wide_range_draws_nothing():
    rs = np.random.RandomState(3)
    mt = orc.MT(3)
    assert rs.randint(0, 1) == 0 and mt.interval(0) == 0
    assert mt.u32() == int(rs.randint(0, 2**32, dtype=np.uint32))


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert orc.philox4x32(0, 0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert orc.philox4x32(f, f, f, f, f, f) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox4x32(0xa4093822, 0x299f31d0, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_perm6400_is_a_permutation_and_looks_uniform():
    rng = np.random.RandomState(0)
    occupancy = np.zeros(6400)
    K = 640
    trials = 400
    for _ in range(trials):
        rk = rng.randint(0, 2**32, size=8, dtype=np.uint32)
        img = np.array([orc.perm6400(rk, i) for i in range(K)])
        assert len(set(img.tolist())) == K and img.min() >= 0 and img.max() < 6400
        occupancy[img] += 1
    rk = rng.randint(0, 2**32, size=8, dtype=np.uint32)
    full = sorted(orc.perm6400(rk, i) for i in range(6400))
    assert full == list(range(6400))
    # each cell is hit with p = K/6400 = 0.1: chi-square over 6400 cells, dof 6399
    expect = trials * K / 6400.0
    chi2 = ((occupancy - expect) ** 2 / (expect * (1 - K / 6400.0))).sum()
    assert abs(chi2 - 6400) < 5 * np.sqrt(2 * 6400), chi2
