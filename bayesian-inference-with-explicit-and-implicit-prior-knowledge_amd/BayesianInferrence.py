"""Matrix-normal-inverse-Wishart algebra (mirror of reference src/BayesianInferrence.py:11-61).

Setup-time functions take and return NumPy arrays, as the reference's drivers call them on host
data (src/Toy_Example.py:38-43).  The per-Gibbs-iteration use inside PGAS.sample_params runs on
torch tensors on the engine's device (see PGAS.py).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla


def _solve_spd(A, B):
    return sla.cho_solve(sla.cho_factor(A, lower=True), B)


def prior_mniw_2naturalPara(mean, col_cov, row_scale, df):
    """(mean, col_cov, row_scale, df) -> (eta_0 (M,n), eta_1 (M,M), eta_2 (n,n), eta_3)  [BI:18-32]."""
    mean = np.atleast_2d(np.asarray(mean, dtype=np.float64))
    row_scale = np.atleast_2d(np.asarray(row_scale, dtype=np.float64))
    col_cov = np.asarray(col_cov, dtype=np.float64)
    n, M = mean.shape
    sol = _solve_spd(col_cov, np.hstack([mean.T, np.eye(M)]))
    eta_0, eta_1 = sol[:, :n], sol[:, n:]
    return eta_0, eta_1, mean @ eta_0 + row_scale, df


def prior_mniw_2naturalPara_inv(eta_0, eta_1, eta_2, eta_3):
    """Inverse map [BI:35-45]."""
    eta_0 = np.asarray(eta_0, dtype=np.float64)
    eta_1 = np.asarray(eta_1, dtype=np.float64)
    n = eta_0.shape[1]
    sol = _solve_spd(eta_1, np.hstack([eta_0, np.eye(eta_1.shape[0])]))
    mean = sol[:, :n].T
    return np.atleast_2d(mean), sol[:, n:], np.atleast_2d(np.asarray(eta_2) - mean @ eta_0), eta_3


def prior_mniw_mean(eta_0, eta_1):
    """Posterior mean with symmetrised eta_1 [BI:48-50]."""
    eta_1 = np.asarray(eta_1, dtype=np.float64)
    return _solve_spd(0.5 * (eta_1 + eta_1.T), np.asarray(eta_0, dtype=np.float64)).T


def prior_mniw_calcStatistics(y, basis):
    """Per-sample sufficient statistics [BI:53-61]."""
    y = np.atleast_1d(np.asarray(y, dtype=np.float64))
    basis = np.atleast_1d(np.asarray(basis, dtype=np.float64))
    return np.outer(basis, y), np.outer(basis, basis), np.outer(y, y), 1


def prior_mniw_Predictive(mean, col_cov, row_scale, df, basis):
    """Matrix-t predictive parameters at `basis` (n_b, M) [BI:64-89]: (mean, col_scale, row_scale / df', df' = df + 1 - n)."""
    basis = np.atleast_2d(np.asarray(basis, dtype=np.float64))
    col_cov = np.atleast_2d(np.asarray(col_cov, dtype=np.float64))
    row_scale = np.atleast_2d(np.asarray(row_scale, dtype=np.float64))
    df = df + 1 - row_scale.shape[0]
    pred_mean = np.squeeze(basis @ np.atleast_2d(np.asarray(mean, dtype=np.float64)).T)
    col_scale = basis @ col_cov @ basis.T + np.eye(basis.shape[0])
    return pred_mean, col_scale, row_scale / df, df


def prior_mniw_drawPred(key, mean, col_scale, row_scale, df):
    """One draw from the matrix-t predictive [BI:92-108]: mean + chol(row) t chol(col)^T with t ~ Student-t(df), one variate per
    row dimension.  `key` is a pgas_amd.random key (own Philox streams: the reference's jax.random.t stream is not reproducible
    here).  The variates are the library's (pgas_m_rng_student_t_host: z sqrt(a / Gamma(a)), a = df / 2, stream 32 of the key) -- the
    arithmetic the device kernels use, so a host draw and a device draw from the same counters are the same number."""
    from . import random as prng
    from ._lib import student_t_host

    Lc = np.linalg.cholesky(np.atleast_2d(np.asarray(col_scale, dtype=np.float64)))
    Lr = np.linalg.cholesky(np.atleast_2d(np.asarray(row_scale, dtype=np.float64)))
    n = Lr.shape[0]
    t = student_t_host(prng.as_key(key), prng.STREAM_INTVAR, 0, np.full(n, float(df)))
    return np.asarray(mean, dtype=np.float64) + np.squeeze(np.einsum("ij,j,jk->ik", Lr, t, Lc.T))


def prior_mniw_log_base_measure(T_0, T_1, T_2, T_3):
    """Log base measure of the MNIW family in natural parameters [BI:111-124]."""
    from scipy.special import multigammaln

    T_0 = np.asarray(T_0, dtype=np.float64)
    T_1 = np.asarray(T_1, dtype=np.float64)
    T_2 = np.atleast_2d(np.asarray(T_2, dtype=np.float64))
    n, m = T_2.shape[0], T_1.shape[0]
    Psi = T_2 - T_0.T @ _solve_spd(T_1, T_0)
    nu = T_3
    return (-0.5 * n * m * np.log(2 * np.pi) + 0.5 * n * np.log(np.linalg.det(T_1)) - 0.5 * nu * n * np.log(2)
            - multigammaln(nu / 2, n) + np.log(np.linalg.det(Psi)) * nu / 2)
