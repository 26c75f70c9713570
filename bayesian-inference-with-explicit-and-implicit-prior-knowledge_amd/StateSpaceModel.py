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


class SymbolicStateSpaceModel(StateSpaceModel):
    """The same model with its callables given as a FACTORY over an array namespace -- `model(xp) -> (transition_model, output_model)`, the
    form pgas_amd.experiments writes the reference's models in -- so that they can be traced (pgas_amd.exprs) into register programs
    and run as ONE HIP launch each (pgas_m_expr_eval) instead of the ~45 elementwise torch launches of an RK4 step: transition_mdl,
    output_mdl, draw_state (with the process noise added in the same launch) and log_likelihood (with the Gaussian log-density).
    Algorithm1 / Algorithm3 hand over their device operations with `bind`; unbound, or for a model the tracer does not understand (it
    says so once, as a warning), everything runs through the torch callables exactly like StateSpaceModel."""

    def __init__(self, process_noise, output_noise, model):
        f, g = model(torch)
        super().__init__(process_noise, output_noise, f, g)
        self._model, self._ops, self._progs = model, None, {}

    def bind(self, ops):
        self._ops = ops

    def _program(self, which, state, input, ivs):
        if self._ops is None:
            return None
        key = (which, state.shape[1], int(input.numel()), tuple(int(v.reshape(v.shape[0], -1).shape[1]) for v in ivs))
        if key not in self._progs:
            from . import exprs

            def traced(st, u, *iv):
                return self._model(exprs.SymNamespace(st.tr))[which](st, u, *iv)

            try:
                self._progs[key] = exprs.trace(traced, key[1], key[2], key[3])
            except TypeError as e:
                import warnings
                warnings.warn(f"the model's {'output' if which else 'transition'} function stays in torch ({e})", RuntimeWarning, stacklevel=3)
                self._progs[key] = None
        return self._progs[key]

    @staticmethod
    def _vec(x, like):
        return torch.as_tensor(x, dtype=torch.float64, device=like.device).reshape(-1)

    def transition_mdl(self, state, input, *int_variables):
        u = self._vec(input, state)
        prog = self._program(0, state, u, int_variables)
        return super().transition_mdl(state, input, *int_variables) if prog is None else self._ops.expr_eval(prog, state, u, int_variables)

    def output_mdl(self, state, input, *int_variables):
        u = self._vec(input, state)
        prog = self._program(1, state, u, int_variables)
        return super().output_mdl(state, input, *int_variables) if prog is None else self._ops.expr_eval(prog, state, u, int_variables)

    def draw_state(self, std_normal, state, input, *int_variables):
        u = self._vec(input, state)
        prog = self._program(0, state, u, int_variables)
        if prog is None:
            return super().draw_state(std_normal, state, input, *int_variables)
        if self.is_deterministic:
            return self._ops.expr_eval(prog, state, u, int_variables)
        return self._ops.expr_eval(prog, state, u, int_variables, mode=1, aux=std_normal, mat=self._dev("Qc", self._Q_chol, state))

    def draw_state_gather(self, std_normal, state, input, anc, *int_variables):
        """draw_state(std_normal, state[anc], input, *[v[anc] ...]) with the gather done by the kernel."""
        u = self._vec(input, state)
        prog = self._program(0, state, u, int_variables)
        if prog is None:
            ai = anc.long()
            return super().draw_state(std_normal, state[ai], input, *[v[ai] for v in int_variables])
        if self.is_deterministic:
            return self._ops.expr_eval(prog, state, u, int_variables, anc=anc)
        return self._ops.expr_eval(prog, state, u, int_variables, mode=1, anc=anc, aux=std_normal, mat=self._dev("Qc", self._Q_chol, state))

    def transition_logpdf(self, target, state, input, *int_variables):
        """log N(target; transition_model(state, input, ...), process_noise) per particle in one launch (src/Algorithm3.py:109-116), or None
        when the transition is deterministic / not traced (the caller then uses its torch expression)."""
        if self.is_deterministic:
            return None
        u = self._vec(input, state)
        prog = self._program(0, state, u, int_variables)
        if prog is None:
            return None
        Lq = self._Q_chol
        cQ = -0.5 * Lq.shape[0] * math.log(2 * math.pi) - float(np.sum(np.log(np.diag(Lq))))
        return self._ops.expr_eval(prog, state, u, int_variables, mode=2, aux=self._vec(target, state), mat=self._dev("LQinv", np.linalg.inv(Lq), state), cR=cQ)

    def log_likelihood(self, observation, state, input, *int_variables):
        u = self._vec(input, state)
        prog = self._program(1, state, u, int_variables)
        if prog is None or len(prog.out_regs) != self.output_noise.shape[0]:
            return super().log_likelihood(observation, state, input, *int_variables)
        return self._ops.expr_eval(prog, state, u, int_variables, mode=2, aux=self._vec(observation, state), mat=self._dev("LRinv", self._LRinv, state), cR=self._cR)
