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
STREAM_INTVAR = 32   # PGAS_STREAM_M_INTVAR (include/pgas_canon.h): Student-t variates of prior_mniw_drawPred

# Normal, chi^2 and Student-t variates are NOT generated in this module: they come from the library (device kernels k_rng_normal /
# k_rng_chi2 / k_rng_student_t, or pgas_m_rng_student_t_host for host-side helpers), one arithmetic for one key type.


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
