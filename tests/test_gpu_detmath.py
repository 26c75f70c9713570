"""The arithmetic primitives shared by the HIP kernels and the canonical C oracle (include/pgas_detmath.h, include/pgas_canon.h),
checked ON THE DEVICE: bit for bit against the host build of the same header, and against mpmath (< 1 ulp) directly -- so an error in
the shared header cannot hide behind "device == oracle"."""
import numpy as np
import pytest

from oracle import canon

pytestmark = pytest.mark.gpu


def _dev(which, **kw):
    from pgas_amd import _lib

    return _lib.detmath_eval(which, **kw)


def _max_ulp(got, fn, xs):
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 200
    worst = 0.0
    for x, g in zip(xs, got):
        r = fn(mp, mp.mpf(float(x)))
        if r == 0:
            assert g == 0
            continue
        worst = max(worst, float(abs(mp.mpf(float(g)) - r) / mp.mpf(float(np.spacing(abs(float(r)))))))
    return worst


def test_exp_log_sincospi_on_device():
    rng = np.random.default_rng(10)
    xe = np.concatenate([rng.uniform(-708, 709, 60000), rng.uniform(-1, 1, 60000), rng.uniform(-40, 0, 60000), [-709.0, -708.0, 0.0, 710.0, -np.inf, np.nan]])
    ge = _dev(0, x=xe)[0]
    assert np.array_equal(ge, canon.det_exp(xe), equal_nan=True), "device exp != host exp"
    assert _max_ulp(ge[:400], lambda mp, v: mp.exp(v), xe[:400]) < 1.0 and _max_ulp(ge[60000:60400], lambda mp, v: mp.exp(v), xe[60000:60400]) < 1.0
    xl = np.concatenate([rng.uniform(0, 1, 60000), 2.0 ** -rng.uniform(0, 53, 60000), rng.uniform(0.5, 2, 60000), [1.0, 2.0 ** -53]])
    gl = _dev(1, x=xl)[0]
    assert np.array_equal(gl, canon.det_log(xl)), "device log != host log"
    assert _max_ulp(gl[:400], lambda mp, v: mp.log(v), xl[:400]) < 1.0 and _max_ulp(gl[60000:60400], lambda mp, v: mp.log(v), xl[60000:60400]) < 1.0
    xs = np.concatenate([rng.uniform(-0.25, 0.25, 60000), rng.uniform(-4, 4, 60000), rng.uniform(-1000, 1000, 60000), [0.0, 0.5, 1.0, 1.5, 2.0, -1.0, 7.0]])
    s, c, _ = _dev(2, x=xs)
    so, co = canon.det_sincospi(xs)
    assert np.array_equal(s, so) and np.array_equal(c, co), "device sincospi != host sincospi"
    assert _max_ulp(s[60000:60400], lambda mp, v: mp.sin(mp.pi * v), xs[60000:60400]) < 1.0
    assert _max_ulp(c[60000:60400], lambda mp, v: mp.cos(mp.pi * v), xs[60000:60400]) < 1.0
    assert np.array_equal(np.abs(s[-7:]), [0, 1, 0, 1, 0, 0, 0]) and np.array_equal(np.abs(c[-7:]), [1, 0, 1, 0, 1, 1, 1])


def test_philox_and_normals_on_device():
    # Random123 kat_vectors (philox4x32, 10 rounds), evaluated by the device
    w = np.array([[0, 0, 0, 0, 0, 0], [0xFFFFFFFF] * 6, [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0]], dtype=np.uint32)
    out = _dev(3, words=w)[2]
    assert out.tolist() == [[0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8], [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD],
                            [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]]
    rng = np.random.default_rng(11)
    w = rng.integers(0, 2**32, (50000, 6), dtype=np.uint64).astype(np.uint32)
    out = _dev(3, words=w)[2]
    ref = np.array([canon.philox(list(map(int, r[:4])), list(map(int, r[4:]))) for r in w[:3000]], dtype=np.uint32)
    assert np.array_equal(out[:3000], ref)
    # Box-Muller pairs: device == host for the engine's own counter layout (seed, stream, t, particle)
    seed, t, n = 987654321, 7, 4096
    words = np.array([[p, 0, t, canon.STREAM_PROP, seed & 0xFFFFFFFF, seed >> 32] for p in range(n)], dtype=np.uint32)
    z0, z1, _ = _dev(7, words=words)
    zo = canon.normals(seed, canon.STREAM_PROP, t, 0, n, 2)
    assert np.array_equal(z0, zo[:, 0]) and np.array_equal(z1, zo[:, 1])


def test_level_references_on_device():
    """pgas_seg_ref / pgas_seg_arg / pgas_lvl_scale (include/pgas_canon.h): closed forms, checked on the device against exact arithmetic."""
    from fractions import Fraction

    rng = np.random.default_rng(12)
    m = np.concatenate([rng.uniform(-800, 50, 5000), [-np.inf, 0.0, 1e-300, -1e-300]])
    kref = _dev(4, x=m)[0]
    fin = np.isfinite(m)
    assert np.all(kref[fin] == np.ceil(m[fin] * float.fromhex("0x1.71547652b82fep+0"))) and kref[~fin][0] == -np.inf
    lw, k = rng.uniform(-700, 0, 4000), np.ceil(rng.uniform(-1000, 0, 4000))
    arg = _dev(5, x=lw, y=k)[0]
    hi, lo = float.fromhex("0x1.62e42fefa39efp-1"), float.fromhex("0x1.abc9e3b39803fp-56")
    for i in range(0, 4000, 40):   # two correctly rounded fused steps, exactly
        step1 = float(Fraction(lw[i]) - Fraction(k[i]) * Fraction(hi))
        step2 = float(Fraction(step1) - Fraction(k[i]) * Fraction(lo))
        assert arg[i] == step2
    kk = np.concatenate([np.ceil(rng.uniform(-600, 0, 2000)), [-np.inf, -480.0, -481.0, 0.0, -np.inf]])
    KK = np.concatenate([np.zeros(2000), [0.0, 0.0, 0.0, 0.0, -np.inf]])
    sc = _dev(6, x=kk, y=KK)[0]
    with np.errstate(invalid="ignore"):   # -inf - -inf: the all-empty case, scale 0 by definition
        d = kk - KK
    exp = np.where(d >= -480.0, np.exp2(np.where(np.isfinite(d), d, -np.inf)), 0.0)
    assert np.array_equal(sc, exp)
