"""Shared test helpers: build the two oracles from a pgas_amd Problem and compare index vectors."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import pgas_amd  # noqa: E402,F401
from oracle import canon, pgas_numpy  # noqa: E402
from pgas_amd import experiments  # noqa: E402


def canon_model(problem, N):
    """Canonical C oracle fed with the same tables the HIP engine receives (pgas_amd/_lib.py Engine.__init__)."""
    bm, lik = problem.basis_fcn, problem.likelihood_fcn
    T = problem.T
    y = np.asarray(problem.observations, dtype=np.float64).reshape(T, -1)
    u = np.asarray(problem.inputs, dtype=np.float64).reshape(T, -1)
    return canon.CanonModel(N, T, problem.nx, y.shape[1], u.shape[1], bm.basis.indices, bm.sel, bm.alpha, bm.beta, bm.basis.norm,
                            lik.H, lik.LRinv, lik.cR, y, u)


def numpy_csmc(problem, N):
    """Literal NumPy restatement (oracle/pgas_numpy.py) with vectorised callables built from reference formulas."""
    bm, lik = problem.basis_fcn, problem.likelihood_fcn
    b = bm.basis
    eig = (np.pi * b.indices.astype(np.float64) / b.size) ** 2     # src/BasisFunctions.py:60
    L = b.size / 2

    def basis(state, u):
        state = np.atleast_2d(state)
        v = state if not np.size(u) else np.hstack([state, np.broadcast_to(np.atleast_1d(u), (state.shape[0], np.size(u)))])
        xc = v[:, bm.sel] / bm.div - b.center
        return np.prod(np.sqrt(1 / L) * np.sin(np.sqrt(eig)[None] * (xc[:, None, :] + L)), axis=2)  # :77-80

    def likelihood(obs, state, u):
        return pgas_numpy.mvn_logpdf(np.atleast_1d(obs), np.atleast_2d(state) @ lik.H.T, lik.R)

    T = problem.T
    return pgas_numpy.condSequentialMonteCarlo(
        N, np.asarray(problem.observations).reshape(T, -1), np.asarray(problem.inputs, dtype=np.float64).reshape(T, -1),
        problem.init_state_mean, problem.init_state_cov, likelihood, basis)


def canon_rand(seed, N, T, nx):
    """The Philox-derived random numbers of the canonical sweep, as explicit arrays for the NumPy oracle."""
    z = np.zeros((T, N, nx))
    for t in range(1, T):
        z[t] = canon.normals(seed, canon.STREAM_PROP, t, 0, N, nx)
    return dict(
        z0=canon.normals(seed, canon.STREAM_INIT, 0, 0, N, nx), z=z,
        u_resample=np.array([canon.uniform(seed, canon.STREAM_RESAMPLE, t) for t in range(T)]),
        u_ancestor=np.array([canon.uniform(seed, canon.STREAM_ANCESTOR, t) for t in range(T)]),
        u_final=canon.uniform(seed, canon.STREAM_FINAL, 0),
    )


def index_mismatch_is_tie(a_canon, a_numpy, W_numpy, U, tol=1e-9):
    """True if every differing index is explained by U_i lying within tol of a CDF boundary."""
    bad = np.nonzero(a_canon != a_numpy)[0]
    for i in bad:
        lo, hi = sorted((int(a_canon[i]), int(a_numpy[i])))
        if not np.all(np.abs(W_numpy[lo:hi] - U[i]) < tol):
            return False
    return True


class CanonRand:
    """Random-number provider for oracle/marginal_numpy.py that reproduces the device's Philox streams (include/pgas_canon.h)
    through the canonical C oracle: same (seed, stream, t, particle) addressing as pgas_amd.Algorithm1.DeviceRand."""

    def __init__(self, seed, N):
        self.seed, self.N = int(seed), int(N)

    def normal(self, stream, t, ncol):
        return canon.normals(self.seed, stream, t, 0, self.N, ncol)

    def uniform(self, stream, t):
        return canon.uniform(self.seed, stream, t)

    def student_t(self, stream, t, nu, n=1):
        """(N, n): component j of an n-component interface variable draws from stream + 16 j (pgas_amd.Algorithm1._draw_int_vars)."""
        nu = np.broadcast_to(np.asarray(nu, dtype=np.float64), (self.N,)).copy()
        return np.stack([canon.student_t(self.seed, stream + 16 * j, t, 0, nu) for j in range(n)], axis=1)


def marginal_oracle(problem, N, kind="Algorithm1"):
    """NumPy restatement (oracle/marginal_numpy.py) of Algorithm1 / Algorithm3 for a pgas_amd.experiments.MarginalProblem."""
    from oracle import marginal_numpy as mo

    ssm = problem.ssm(mo.StateSpaceModel, np)
    args = dict(N_samples=N, observations=problem.observations, inputs=problem.inputs, SSM=ssm, init_state_mean=problem.init_state_mean,
                init_state_cov=problem.init_state_cov, init_int_var_mean=problem.init_int_var_mean, init_int_var_cov=problem.init_int_var_cov,
                GP_prior=problem.GP_prior, basis_fcn=problem.basis_fcn())
    if kind == "Algorithm1":
        return mo.Algorithm1(forgetting_factor=problem.forgetting_factor, **args)
    return mo.Algorithm3(**args)


def host_param_draws(key, nx, M, df):
    """The numbers PGAS.param_draws generates on the device, recomputed on the host from the same key by the canonical C oracle:
    split(key) -> (key_A, key_S), split(key_S) -> (key_chi, key_norm); chi^2(df - i), (nx, nx) and (nx, M) standard normals."""
    from pgas_amd import random as prng
    key_A, key_S = prng.split(key, 2)
    key_chi, key_norm = prng.split(key_S, 2)
    return dict(chi2=canon.chi2(key_chi, prng.STREAM_PARAM_UNIFORM, 0, 0, df - np.arange(nx, dtype=np.float64)),
                normals_T=canon.normals(key_norm, prng.STREAM_PARAM_NORMAL, 0, 0, nx * nx, 1).reshape(nx, nx),
                normals_A=canon.normals(key_A, prng.STREAM_PARAM_NORMAL, 0, 0, nx * M, 1).reshape(nx, M))
