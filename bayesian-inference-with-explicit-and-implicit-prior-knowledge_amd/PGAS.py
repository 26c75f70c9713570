"""Host-side mirror of reference src/PGAS.py over the HIP engine.

Same class names, constructor arguments and return conventions as the reference:

* ``condSequentialMonteCarlo(N_samples, observations, inputs, init_state_mean, init_state_cov,
  likelihood_fcn, basis_fcn)`` with ``.step(...)`` and ``__call__(key, ref_state, coeff_mat, error_cov)``
  (src/PGAS.py:24-33, 79-88, 176-182)
* ``PGAS(N_samples, N_iterations, observations, inputs, init_state_mean, init_state_cov,
  likelihood_fcn, GP_prior, basis_fcn)`` with ``sample_params`` and
  ``__call__(key, init_ref_state) -> (state_trace (T,K,nx), log_likelihood (T,K))`` (:237-248, 288-397)

Differences that follow from the boundary (documented in DESIGN.md):
``likelihood_fcn`` is a :class:`GaussianLikelihood`, ``basis_fcn`` a :class:`BasisMap`
(declarative, because a HIP kernel cannot trace Python lambdas); ``key`` is an integer
(``pgas_amd.random.key``); arrays are torch fp64 tensors on the engine's device.
"""
from __future__ import annotations

import numpy as np
import torch

from . import random as prng
from ._lib import Engine
from .descriptors import BasisMap, GaussianLikelihood


class condSequentialMonteCarlo:
    def __init__(self, N_samples, observations, inputs, init_state_mean, init_state_cov, likelihood_fcn, basis_fcn,
                 device=None, keep_logw_trace=False, resample_before_propagate=False):
        """resample_before_propagate=True selects the CORRECTED mode: x_t[i] is drawn from the transition of the resampled
        ancestor x_{t-1}[a_i].  The reference (and the default here) propagates x_{t-1}[i] itself (src/PGAS.py:131-133,
        SURVEY quirk Q1) although it records and weights with a_i; the corrected mode is NOT the reference's behaviour."""
        if not isinstance(likelihood_fcn, GaussianLikelihood):
            raise TypeError("likelihood_fcn must be a pgas_amd.GaussianLikelihood descriptor (HIP kernels cannot trace Python callables)")
        if not isinstance(basis_fcn, BasisMap):
            raise TypeError("basis_fcn must be a pgas_amd.BasisMap descriptor, e.g. basis.on(sel=[0, 1])")
        self.N_samples = int(N_samples)
        self.observations = np.asarray(observations, dtype=np.float64)
        self.inputs = np.asarray(inputs, dtype=np.float64)
        self.init_state_mean = np.asarray(init_state_mean, dtype=np.float64).reshape(-1)
        self.init_state_cov = np.atleast_2d(np.asarray(init_state_cov, dtype=np.float64))
        self.likelihood_fcn = likelihood_fcn
        self.basis_fcn = basis_fcn
        u0 = self.inputs[0] if self.inputs.size else None
        self.dim_basis = len(basis_fcn(self.init_state_mean, u0))  # src/PGAS.py:41-43
        self.engine = Engine(self.N_samples, self.observations, self.inputs, self.init_state_mean, self.init_state_cov,
                             likelihood_fcn, basis_fcn, device=device, keep_logw_trace=keep_logw_trace)
        self.device = self.engine.device
        self.resample_before_propagate = bool(resample_before_propagate)
        if self.resample_before_propagate:
            self.engine.set_option(5, 1)  # PGAS_OPT_RESAMPLE_BEFORE_PROPAGATE

    # src/PGAS.py:45-57
    def _generate_auxiliary_states(self, state, time, coeff_mat, error_cov=None):
        self.engine.set_params(coeff_mat, np.eye(self.engine.nx) if error_cov is None else error_cov)
        return self.engine.aux_states(state, int(time))

    # src/PGAS.py:79-153
    def step(self, key, time, log_weights, state, coeff_mat, error_cov, ref_state):
        """One conditional-SMC step; returns (new_log_weights (N,), new_state (N,nx), a_indices (N,) int32)."""
        self.engine.set_params(coeff_mat, error_cov)
        return self.engine.step(int(time), prng.as_key(key), log_weights, state, ref_state)

    # src/PGAS.py:176-228
    def __call__(self, key, ref_state, coeff_mat, error_cov):
        """Whole sweep on the device; returns the sampled trajectory (T,nx), squeezed like the reference.  With coeff_mat / error_cov
        already on the device (PGAS.sample_params) nothing in here touches the host: error_cov is factored by the pack kernel
        (pgas_set_params_dev) and the sweep itself is one replayed HIP graph."""
        self.engine.set_params(coeff_mat, error_cov)
        ref = ref_state if isinstance(ref_state, torch.Tensor) else torch.as_tensor(np.asarray(ref_state, dtype=np.float64))
        traj = self.engine.sweep(prng.as_key(key), ref.reshape(self.engine.T, self.engine.nx))
        return traj.squeeze(-1) if self.engine.nx == 1 else traj


class PGAS:
    def __init__(self, N_samples, N_iterations, observations, inputs, init_state_mean, init_state_cov, likelihood_fcn,
                 GP_prior, basis_fcn, device=None, resample_before_propagate=False):
        self.N_iterations = int(N_iterations)
        self.N_steps = np.asarray(observations).shape[0]
        self.cSMC = condSequentialMonteCarlo(N_samples, observations, inputs, init_state_mean, init_state_cov,
                                             likelihood_fcn, basis_fcn, device=device,
                                             resample_before_propagate=resample_before_propagate)
        dev = self.cSMC.device
        f = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)  # noqa: E731
        self.GP_prior = (f(GP_prior[0]), f(GP_prior[1]), f(np.atleast_2d(GP_prior[2])), float(GP_prior[3]))
        self._df_vec = None
        self._eye = (torch.eye(self.cSMC.engine.M, dtype=torch.float64, device=dev), torch.eye(self.cSMC.engine.nx, dtype=torch.float64, device=dev))

    # ---- src/PGAS.py:288-343 ------------------------------------------------------------------
    def param_draws(self, key):
        """The random numbers sample_params consumes: chi^2(df - i) (:323-327), tril normals (:328), A normals (:338) -- drawn on the
        device from the Philox streams of the three sub-keys (k_rng_chi2, k_rng_normal); nothing crosses PCIe."""
        eng = self.cSMC.engine
        nx, M = eng.nx, eng.M
        df = self.GP_prior[3] + (self.N_steps - 1)
        key_A, key_S = prng.split(key, 2)
        key_chi, key_norm = prng.split(key_S, 2)
        if self._df_vec is None or self._df_vec[0] != df:
            self._df_vec = (df, torch.as_tensor(df - np.arange(nx, dtype=np.float64), device=eng.device))
        return dict(chi2=eng.rng_chi2(key_chi, prng.STREAM_PARAM_UNIFORM, 0, self._df_vec[1]),
                    normals_T=eng.rng_normal(key_norm, prng.STREAM_PARAM_NORMAL, 0, nx * nx).reshape(nx, nx),
                    normals_A=eng.rng_normal(key_A, prng.STREAM_PARAM_NORMAL, 0, nx * M).reshape(nx, M))

    def sample_params(self, key, state_trajectory, draws=None):
        """(A (nx,M), S (nx,nx)) ~ p(A, S | trajectory): suff-stats on the engine (fp64 MFMA SYRK),
        MNIW algebra with torch.linalg on the same device."""
        eng = self.cSMC.engine
        dev = eng.device
        T0, T1, T2, T3 = eng.suffstats(state_trajectory)                      # :294-303
        e0, e1, e2, e3 = self.GP_prior[0] + T0, self.GP_prior[1] + T1, self.GP_prior[2] + T2, self.GP_prior[3] + T3
        # cholesky_ex: no host-side error check, i.e. no synchronisation inside a Gibbs iteration (a failed factorisation shows up as NaNs)
        Lc = torch.linalg.cholesky_ex(e1, check_errors=False)[0]               # BI:35-45
        sol = torch.cholesky_solve(torch.cat([e0, self._eye[0]], dim=1), Lc)
        mean, col_cov = sol[:, : eng.nx].T.contiguous(), sol[:, eng.nx:]
        row_scale = e2 - mean @ e0
        if draws is None:
            draws = self.param_draws(key)
        g = lambda a: a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)  # noqa: E731
        eye = self._eye[1]
        L = torch.linalg.solve_triangular(torch.linalg.cholesky_ex(row_scale, check_errors=False)[0], eye, upper=False)      # :317-319
        Tm = torch.tril(g(draws["normals_T"]), diagonal=-1) + torch.diag(torch.sqrt(g(draws["chi2"])))  # :327-329
        Cm = L @ Tm                                                            # :332
        S_chol = torch.linalg.solve_triangular(Cm.T.contiguous(), eye, upper=True)  # :334
        S = S_chol @ S_chol.T                                                  # :335
        V_chol = torch.linalg.cholesky_ex(col_cov, check_errors=False)[0]      # :339
        A = mean + S_chol @ g(draws["normals_A"]) @ V_chol                     # :341 (Q5)
        self.last_df = e3
        return A, S

    # ---- src/PGAS.py:345-397 ------------------------------------------------------------------
    def __call__(self, key, init_ref_state, progress=None):
        eng = self.cSMC.engine
        dev = eng.device
        K, T, nx = self.N_iterations, self.N_steps, eng.nx
        state_trace = torch.zeros((K, T, nx), dtype=torch.float64, device=dev)            # :266-273
        state_trace[0] = torch.as_tensor(np.asarray(init_ref_state, dtype=np.float64), device=dev).reshape(T, nx)
        key, key_para = prng.split(key, 2)                                                # :356
        coeff_mat, error_cov = self.sample_params(key_para, state_trace[0])               # :358
        # what the chain consumed, iteration by iteration (not in the reference; the parity test replays the chain from it)
        self.chain_log = dict(step_keys=[None], para_keys=[key_para], params=[(coeff_mat, error_cov)])
        for k in range(1, K):                                                             # :361
            key, key_step = prng.split(key, 2)                                            # :365
            new_state = self.cSMC(key_step, state_trace[k - 1], coeff_mat, error_cov)     # :366-371
            state_trace[k] = new_state.reshape(T, nx)                                     # :374
            key, key_para = prng.split(key, 2)                                            # :377
            coeff_mat, error_cov = self.sample_params(key_para, state_trace[k])           # :378
            self.chain_log["step_keys"].append(key_step)
            self.chain_log["para_keys"].append(key_para)
            self.chain_log["params"].append((coeff_mat, error_cov))
            if progress is not None:
                progress(k)
        state_trace = state_trace.transpose(0, 1).contiguous()                            # :380 -> (T,K,nx)
        # :383-392  log N(y_t; H x_tk, R) for every (t, k)
        lik = self.cSMC.likelihood_fcn
        y = torch.as_tensor(self.cSMC.observations.reshape(T, -1), device=dev)
        e = y[:, None, :] - state_trace @ torch.as_tensor(lik.H, device=dev).T
        w = e @ torch.as_tensor(lik.LRinv, device=dev).T
        log_likelihood = lik.cR - 0.5 * (w * w).sum(-1)
        self.coeff_mat, self.error_cov = coeff_mat, error_cov
        return state_trace, log_likelihood
