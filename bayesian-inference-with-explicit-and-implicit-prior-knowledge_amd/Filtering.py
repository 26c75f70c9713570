"""Mirror of reference src/Filtering.py on the device.

``systematic_SISR(key, w)`` (src/Filtering.py:6-37) and ``reconstruct_trajectory(Particles, ancestry, idx)`` (:40-55) with the
reference's signatures; arrays may be NumPy or torch, results are torch tensors on the GPU.  Both run hand-written HIP
kernels through the C ABI (pgas_systematic_resample / pgas_reconstruct_trajectory); there is no CPU path.
"""
from __future__ import annotations

import numpy as np
import torch

from . import random as prng
from ._lib import Engine

STREAM_SISR = 32


def systematic_SISR(key, w, device=None):
    """Systematic resampling indices (N,) int32 for (unnormalised) weights w.

    `key` is an integer key; the uniform of src/Filtering.py:19 is drawn from its Philox stream.  Negative weights are
    clipped to zero (:23); if no weight is positive the identity map is returned (:25).
    """
    wt = w if isinstance(w, torch.Tensor) else torch.as_tensor(np.asarray(w, dtype=np.float64))
    N = wt.numel()
    eng = Engine.utility(N, device)
    wt = wt.to(device=eng.device, dtype=torch.float64).reshape(N)
    u = float(prng.uniform(prng.as_key(key), 1, stream=STREAM_SISR)[0])
    logw = torch.log(torch.clamp(wt, min=0.0))   # log 0 = -inf: such particles get no offspring
    return eng.systematic_resample(u, logw)


def reconstruct_trajectory(Particles, ancestry, idx, device=None):
    """Back-trace (src/Filtering.py:40-55): traj[T-1] = P[T-1, idx], then follow ancestry backwards; squeezed like the reference."""
    P = Particles if isinstance(Particles, torch.Tensor) else torch.as_tensor(np.asarray(Particles, dtype=np.float64))
    N = P.shape[1]
    traj = Engine.utility(N, device).reconstruct_trajectory(P, ancestry, idx)
    return traj.squeeze()
