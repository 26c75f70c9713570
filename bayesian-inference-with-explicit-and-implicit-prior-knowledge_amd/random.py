"""Counter-based keys for the engine.

The reference threads ``jax.random.key`` objects through its loops (src/PGAS.py:184,203,311,356,365);
JAX's threefry streams cannot be regenerated offline, so this package uses its own Philox4x32-10
streams (include/pgas_detmath.h) and keeps only the *structure* of the reference's key handling:
``key(seed)``, ``split(key, n)``.  A key is a plain 64-bit integer.
"""
from __future__ import annotations

import numpy as np

_M0, _M1 = 0xD2511F53, 0xCD9E8D57
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = 0xFFFFFFFF

STREAM_SPLIT = 16
STREAM_PARAM_NORMAL = 17
STREAM_PARAM_UNIFORM = 18


def philox4x32_10(ctr, key):
    """Philox4x32-10 on Python ints; ctr = 4 words, key = 2 words -> 4 words."""
    c0, c1, c2, c3 = (int(v) & _MASK for v in ctr)
    k0, k1 = (int(v) & _MASK for v in key)
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & _MASK, p1 & _MASK, ((p0 >> 32) ^ c3 ^ k1) & _MASK, p0 & _MASK
        k0, k1 = (k0 + _W0) & _MASK, (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def key(seed: int) -> int:
    return int(seed) & 0xFFFFFFFFFFFFFFFF


def as_key(k) -> int:
    if isinstance(k, (int, np.integer)):
        return key(int(k))
    if hasattr(k, "seed"):
        return key(int(k.seed))
    raise TypeError(f"expected an integer key, got {type(k).__name__}")


def split(k, num: int = 2):
    """Derive `num` child keys (the analogue of jax.random.split)."""
    k = as_key(k)
    out = []
    for i in range(num):
        w = philox4x32_10((i, 0, 0, STREAM_SPLIT), (k & _MASK, k >> 32))
        out.append(w[0] | (w[1] << 32))
    return out


def _u52(lo, hi):
    return (((hi << 32) | lo) >> 12) * 2.0**-52 + 2.0**-53


def uniform(k, n: int, stream: int = STREAM_PARAM_UNIFORM) -> np.ndarray:
    k = as_key(k)
    out = np.empty(n)
    for i in range(0, n, 2):
        w = philox4x32_10((i // 2, 0, 0, stream), (k & _MASK, k >> 32))
        out[i] = _u52(w[0], w[1])
        if i + 1 < n:
            out[i + 1] = _u52(w[2], w[3])
    return out


def normal(k, shape, stream: int = STREAM_PARAM_NORMAL) -> np.ndarray:
    """Standard normals (Box-Muller on Philox uniforms); used for parameter draws only."""
    k = as_key(k)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    out = np.empty(n + (n & 1))
    for i in range(0, n, 2):
        w = philox4x32_10((i // 2, 0, 0, stream), (k & _MASK, k >> 32))
        ua, ub = _u52(w[0], w[1]), _u52(w[2], w[3])
        rad = np.sqrt(-2.0 * np.log(ua))
        out[i], out[i + 1] = rad * np.cos(2 * np.pi * ub), rad * np.sin(2 * np.pi * ub)
    return out[:n].reshape(shape)


def chisquare(k, df) -> np.ndarray:
    """chi^2(df_i) draws via Marsaglia-Tsang gamma sampling on Philox streams (df_i > 0)."""
    df = np.atleast_1d(np.asarray(df, dtype=np.float64))
    out = np.empty_like(df)
    subkeys = split(k, len(df))
    for i, (nu, sk) in enumerate(zip(df, subkeys)):
        a = nu / 2.0
        boost = 1.0
        if a < 1.0:  # Gamma(a) = Gamma(a+1) * U^(1/a)
            boost = uniform(sk, 1, stream=STREAM_PARAM_UNIFORM + 1)[0] ** (1.0 / a)
            a += 1.0
        d = a - 1.0 / 3.0
        c = 1.0 / np.sqrt(9.0 * d)
        zs = normal(sk, 64)
        us = uniform(sk, 64)
        val = None
        for z, u in zip(zs, us):
            v = (1.0 + c * z) ** 3
            if v > 0 and np.log(u) < 0.5 * z * z + d - d * v + d * np.log(v):
                val = d * v
                break
        if val is None:  # 64 consecutive rejections has probability < 1e-80
            raise RuntimeError("gamma sampler failed to accept")
        out[i] = 2.0 * val * boost
    return out
