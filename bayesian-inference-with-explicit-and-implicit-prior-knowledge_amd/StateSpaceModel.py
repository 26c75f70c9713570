"""Mirror of reference src/StateSpaceModel.py:8-87 on torch-ROCm tensors.

The reference evaluates `transition_model(state, input, *int_var)` / `output_model(...)` per particle under jax.vmap.  Here
the same callables are written once for the whole batch: `state` is (N, n_x), `input` is (n_u,), every interface variable is
(N, n_i); they return (N, n_x) / (N, n_y) (or (N,) for a scalar output) and must be built from torch operations so they run
on the device.  Everything else -- names, argument order, `is_deterministic`, the Gaussian likelihood -- follows the reference.
"""
from __future__ import annotations

import math

import numpy as np
import torch


class StateSpaceModel:
    def __init__(self, process_noise, output_noise, transition_model, output_model):
        self.process_noise = np.atleast_2d(np.asarray(process_noise, dtype=np.float64))
        self.output_noise = np.atleast_2d(np.asarray(output_noise, dtype=np.float64))
        self.transition_model = transition_model
        self.output_model = output_model
        self.is_deterministic = bool(np.all(self.process_noise == 0))            # :30
        self._Q_chol = None if self.is_deterministic else np.linalg.cholesky(self.process_noise)
        Lr = np.linalg.cholesky(self.output_noise)
        self._LRinv = np.linalg.inv(Lr)
        ny = self.output_noise.shape[0]
        self._cR = -0.5 * ny * math.log(2 * math.pi) - float(np.sum(np.log(np.diag(Lr))))
        self._cache = {}

    def _dev(self, name, arr, like):
        key = (name, like.device)
        if key not in self._cache:
            self._cache[key] = torch.as_tensor(arr, dtype=torch.float64, device=like.device)
        return self._cache[key]

    def transition_mdl(self, state, input, *int_variables):                       # :32-42
        return self.transition_model(state, input, *int_variables)

    def output_mdl(self, state, input, *int_variables):                           # :44-54
        return self.output_model(state, input, *int_variables)

    def draw_state(self, std_normal, state, input, *int_variables):               # :56-74
        """`std_normal` (N, n_x): the standard normals the reference draws from its key at :67."""
        new_state = self.transition_mdl(state, input, *int_variables)
        if self.is_deterministic:
            return new_state
        return new_state + std_normal @ self._dev("Qc", self._Q_chol, state).T

    def log_likelihood(self, observation, state, input, *int_variables):          # :76-87
        out = self.output_mdl(state, input, *int_variables)
        out = out.reshape(out.shape[0], -1)
        e = (torch.as_tensor(observation, dtype=torch.float64, device=out.device).reshape(1, -1) - out) @ self._dev("LRinv", self._LRinv, out).T
        return self._cR - 0.5 * (e * e).sum(dim=1)
