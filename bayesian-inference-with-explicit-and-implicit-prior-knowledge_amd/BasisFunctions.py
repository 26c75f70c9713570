"""Host-side setup of the Hilbert-space GP basis (mirror of reference src/BasisFunctions.py)."""
from __future__ import annotations

import heapq

import numpy as np

from .descriptors import HilbertBasis


def _select_indices(num_fcn, size, idx_start, idx_step):
    """The num_fcn index tuples with the smallest Laplacian eigenvalue sum_d (pi j_d / size_d)^2.

    Must reproduce the reference's ordering exactly (it defines the layout of A, T0, T1):
    best-first expansion from the smallest corner, priority = (running float cost, lattice position),
    neighbour cost = parent cost + weight_d * (j_new^2 - j_old^2)  (src/BasisFunctions.py:24-57).
    """
    D = len(size)
    freqs = np.arange(idx_start, num_fcn * idx_step + 1 + idx_start, idx_step)
    sq = freqs**2
    wgt = (np.pi / size) ** 2
    start = (0,) * D
    queue = [(float(np.sum(wgt * sq[0])), start)]
    pushed = {start}
    picked = []
    while queue and len(picked) < num_fcn:
        cost, at = heapq.heappop(queue)
        picked.append([int(freqs[i]) for i in at])
        for d in range(D):
            step_to = at[d] + 1
            if step_to >= len(freqs):
                continue
            nb = at[:d] + (step_to,) + at[d + 1 :]
            if nb not in pushed:
                pushed.add(nb)
                heapq.heappush(queue, (cost + float(wgt[d] * (sq[step_to] - sq[at[d]])), nb))
    return np.asarray(picked, dtype=np.int32)


def spectral_density_Gaussian(freq, magnitude, lengthscale):
    """Spectral density of the squared-exponential kernel (src/BasisFunctions.py:83-105)."""
    freq = np.asarray(freq, dtype=np.float64)
    ls = np.broadcast_to(lengthscale, freq.shape)
    return magnitude * (2 * np.pi) ** (len(freq) / 2) * np.prod(ls) * np.exp(-0.5 * np.sum(ls**2 * freq**2))


def generate_Hilbert_BasisFunction(num_fcn, domain_boundary, lengthscale, scale, idx_start=1, idx_step=1):
    """Same signature and return convention as src/BasisFunctions.py:8-74: ``(basis, spectral_density)``.

    ``basis`` is a :class:`HilbertBasis` (callable like the reference's jitted closure, and usable
    as an engine descriptor via ``basis.on(sel, div)``).
    """
    box = np.atleast_2d(np.asarray(domain_boundary, dtype=np.float64))
    if idx_start < 1:
        idx_start = 1
    size = box[:, 1] - box[:, 0]
    basis = HilbertBasis(_select_indices(num_fcn, size, idx_start, idx_step), box)
    sd = np.array([spectral_density_Gaussian(f, scale, lengthscale) for f in np.sqrt(basis.eigenvalues)])
    return basis, sd
