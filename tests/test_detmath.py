"""include/pgas_detmath.h: accuracy against mpmath and the Philox4x32-10 known-answer vectors."""
import numpy as np
import pytest

from oracle import canon

mp = pytest.importorskip("mpmath")
mp.mp.prec = 200


def _max_ulp(got, fn, xs):
    worst = 0.0
    for x, g in zip(xs, got):
        r = fn(mp.mpf(float(x)))
        if r == 0:
            assert g == 0
            continue
        worst = max(worst, float(abs(mp.mpf(float(g)) - r) / mp.mpf(float(np.spacing(abs(float(r)))))))
    return worst


def test_philox_random123_kat():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert canon.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert canon.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert canon.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_host_philox_matches_c():
    from pgas_amd import random as prng

    for ctr, key in (([1, 2, 3, 4], [5, 6]), ([0xFFFFFFFF, 0, 7, 0x80000000], [0xDEADBEEF, 0x12345678])):
        assert list(prng.philox4x32_10(ctr, key)) == canon.philox(ctr, key)


def test_exp_accuracy_and_edges():
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-708, 709, 1500), rng.uniform(-1, 1, 1500), rng.uniform(-40, 0, 1500)])
    assert _max_ulp(canon.det_exp(xs), mp.exp, xs) < 1.0
    e = canon.det_exp(np.array([-709.0, -708.0, 0.0, 710.0, -np.inf]))
    assert e[0] == 0.0 and e[1] > 0 and e[2] == 1.0 and np.isinf(e[3]) and e[4] == 0.0
    assert np.isnan(canon.det_exp(np.array([np.nan]))[0])


def test_log_accuracy():
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(0, 1, 2000), 2.0 ** -rng.uniform(0, 53, 1500), rng.uniform(0.5, 2, 1000)])
    assert _max_ulp(canon.det_log(xs), mp.log, xs) < 1.0
    assert canon.det_log(np.array([1.0]))[0] == 0.0


def test_sincospi_accuracy_and_exact_zeros():
    rng = np.random.default_rng(2)
    xs = np.concatenate([rng.uniform(-0.25, 0.25, 1500), rng.uniform(-4, 4, 1500), rng.uniform(-1000, 1000, 1000)])
    s, c = canon.det_sincospi(xs)
    assert _max_ulp(s, lambda x: mp.sin(mp.pi * x), xs) < 1.0
    assert _max_ulp(c, lambda x: mp.cos(mp.pi * x), xs) < 1.0
    s, c = canon.det_sincospi(np.array([0.0, 0.5, 1.0, 1.5, 2.0, -1.0, 7.0]))
    assert np.array_equal(np.abs(s), [0, 1, 0, 1, 0, 0, 0])  # Dirichlet zeros on the box boundary are exact
    assert np.array_equal(np.abs(c), [1, 0, 1, 0, 1, 1, 1])


def test_uniform_open_interval_and_normals():
    us = np.array([canon.uniform(s, 3, t) for s in range(40) for t in range(25)])
    assert us.min() > 0 and us.max() < 1
    z = canon.normals(42, canon.STREAM_PROP, 3, 0, 400000, 2)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3 and abs(np.corrcoef(z.T)[0, 1]) < 5e-3
    assert abs((z**4).mean() - 3) < 0.05
    # counter addressing: disjoint particle ranges reproduce the same numbers
    assert np.array_equal(canon.normals(42, canon.STREAM_PROP, 3, 1000, 10, 2), z[1000:1010])
