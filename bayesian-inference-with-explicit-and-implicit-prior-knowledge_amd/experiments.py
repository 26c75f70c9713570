"""Synthetic problem definitions for the BASELINE.json configurations.

These are *configuration sources*: the constants are the reference's (cited per line), the data
are simulated here with NumPy (seed 12345678, the reference's seed value; the stream necessarily
differs from JAX's).  Nothing here is on the timed path.

* SMO-PGAS  (configs 1, 2, 4): the SingleMassOscillator data with a plain-PGAS instantiation
  modelled on src/Toy_Example.py:135-147 / src/EMPS.py:243-255.  The reference never instantiates
  PGAS for this system -- this instantiation is the build's.
* Toy       : src/Toy_Example.py, the only data-free PGAS instantiation in the reference.
* EMPS-PGAS (config 5): src/EMPS.py:101-123,243-255 with synthetic data from the reference's
  linear-friction model (:169-193) because DATA_EMPS.mat is not distributed.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .BasisFunctions import generate_Hilbert_BasisFunction
from .BayesianInferrence import prior_mniw_2naturalPara
from .descriptors import BasisMap, GaussianLikelihood


@dataclass
class Problem:
    name: str
    observations: np.ndarray   # (T,) or (T,ny)
    inputs: np.ndarray         # (T,) / (T,nu) / (T,0)
    init_state_mean: np.ndarray
    init_state_cov: np.ndarray
    likelihood_fcn: GaussianLikelihood
    basis_fcn: BasisMap
    GP_prior: tuple
    X_true: np.ndarray         # (T,nx) simulated truth (initial reference trajectory)

    @property
    def T(self):
        return self.observations.shape[0]

    @property
    def nx(self):
        return self.init_state_mean.shape[0]


# ---------------------------------------------------------------- SingleMassOscillator
def _smo_rk4(x, F, F_sd, dt, m=0.2):
    # src/SingleMassOscillator.py:32-44 (F_sd frozen over the step, as the reference's driver does, :125-126)
    def dx(s):
        return np.array([s[1], (-F_sd + F) / m])

    k1 = dx(x)
    k2 = dx(x + dt / 2.0 * k1)
    k3 = dx(x + dt / 2.0 * k2)
    k4 = dx(x + dt * k3)
    return x + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)


def smo_pgas(T=2000, seed=12345678):
    c1, c2, d1, d2, m = 5.0, 2.0, 0.4, 0.4, 0.2            # src/SingleMassOscillator.py:17-21
    dt = 0.02                                              # :78
    x0, P0 = np.array([0.0, 0.0]), np.diag([1e-4, 1e-4])   # :85-86
    R, Q = np.array([[1e-3]]), np.diag([5e-8, 5e-9])       # :90-91
    t_end = T * dt
    F_ext = np.ones(T) * 9.81 * m                          # :95-97
    F_ext[int(t_end / (3 * dt)):] = 0
    F_ext[int(2 * t_end / (3 * dt)):] = -9.81 * m
    rng = np.random.default_rng(seed)
    X, Y = np.zeros((T, 2)), np.zeros(T)
    X[0] = x0
    Lq = np.linalg.cholesky(Q)
    for i in range(1, T):                                  # :122-130
        F_sd = c1 * X[i - 1, 0] + c2 * X[i - 1, 0] ** 3 + d1 * X[i - 1, 1] / (1 + d2 * X[i - 1, 1] * np.tanh(X[i - 1, 1]))
        X[i] = _smo_rk4(X[i - 1], F_ext[i - 1], F_sd, dt, m) + Lq @ rng.standard_normal(2)
        Y[i] = X[i, 0] + rng.standard_normal() * np.sqrt(R[0, 0])
    M = 41                                                 # :54-60
    basis, sd = generate_Hilbert_BasisFunction(M, np.array([[-7.5, 7.5], [-7.5, 7.5]]), 7.5 * 2 / M, 100)
    prior = prior_mniw_2naturalPara(np.zeros((2, M)), np.diag(sd), np.eye(2), 3)  # n_x = 2 rows (cf. src/EMPS.py:116-123)
    return Problem("SMO-PGAS", Y, F_ext, x0, P0, GaussianLikelihood.of_component(0, 2, R), basis.on([0, 1]), prior, X)


# ---------------------------------------------------------------- Toy example
def toy(T=40, seed=12345678):
    rng = np.random.default_rng(seed)
    Qv, Rv = 4.0, 4.0                                      # src/Toy_Example.py:62-63
    X, Y = np.zeros((T, 1)), np.zeros((T, 1))
    for i in range(1, T):                                  # :88-96; f_x = 10 sinc(x/7) (:18-19), numpy sinc = sin(pi x)/(pi x)
        X[i] = 10 * np.sinc(X[i - 1] / 7) + rng.standard_normal() * np.sqrt(Qv)
        Y[i] = X[i] + rng.standard_normal() * np.sqrt(Rv)
    M = 40                                                 # :29-36
    basis, sd = generate_Hilbert_BasisFunction(M, np.array([-30, 30]), 3, 50)
    prior = prior_mniw_2naturalPara(np.zeros((1, M)), np.diag(sd), np.eye(1), 10)  # :38-43
    return Problem("Toy", Y, np.zeros((T, 0)), np.array([0.0]), np.diag([1e-4]), GaussianLikelihood(np.eye(1), np.diag([Rv])),
                   basis.on([0]), prior, X)


# ---------------------------------------------------------------- EMPS (synthetic data)
def emps_pgas(T=2000, seed=12345678, M=729):
    dt = 0.01                                              # 1 kHz data decimated x10, src/EMPS.py:59-65
    rng = np.random.default_rng(seed)

    def dx(s, tau):                                        # src/EMPS.py:169-173
        return np.array([s[1], (tau - 203.5 * s[1] - 20.39 * np.sign(s[1]) + 3.16) / 95.11])

    # trapezoidal bang-bang force keeping |q| < 0.4, |dq| < 0.4, |tau| < 160
    tt = np.arange(T) * dt
    tau = 60.0 * np.sign(np.sin(2 * np.pi * tt / 4.0)) * np.minimum(1.0, 4 * np.abs(np.sin(2 * np.pi * tt / 4.0)))
    X = np.zeros((T, 2))
    for i in range(1, T):                                  # RK4, :186-193
        s, u = X[i - 1], tau[i - 1]
        k1 = dx(s, u); k2 = dx(s + dt * k1 / 2, u); k3 = dx(s + dt * k2 / 2, u); k4 = dx(s + dt * k3, u)
        X[i] = s + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    R = np.diag([1e-4])                                    # :74
    Y = X[:, 0] + rng.standard_normal(T) * 1e-2
    x0, P0 = np.array([Y[0], 0.0]), np.diag([1e-5, 1e-6])  # :69-70
    basis, sd = generate_Hilbert_BasisFunction(M, np.array([[-1, 1], [-1, 1], [-1, 1]]), 0.5 / M, 20)  # :101-107
    prior = prior_mniw_2naturalPara(np.zeros((2, M)), np.diag(sd), np.eye(2), 2)                       # :116-123
    bmap = basis.on([0, 1, 2], div=[0.4, 0.4, 160])        # :110-113
    return Problem("EMPS-PGAS", Y, tau, x0, P0, GaussianLikelihood.of_component(0, 2, R), bmap, prior, X)


def initial_params(problem: Problem):
    """A reproducible, well-conditioned (A, S) to run a sweep with: the MNIW posterior given the true
    trajectory (the deterministic part of PGAS.sample_params, src/PGAS.py:294-306): A = posterior mean,
    S = row_scale / df.  Host NumPy, setup only."""
    from .BayesianInferrence import prior_mniw_2naturalPara_inv

    T = problem.T
    u = problem.inputs
    Phi = np.stack([problem.basis_fcn(problem.X_true[t], u[t] if np.size(u) else None) for t in range(T - 1)])
    Xp = problem.X_true[1:]
    e0, e1, e2, e3 = problem.GP_prior
    mean, _, row_scale, df = prior_mniw_2naturalPara_inv(e0 + Phi.T @ Xp, e1 + Phi.T @ Phi, e2 + Xp.T @ Xp, e3 + (T - 1))
    return np.ascontiguousarray(mean), np.atleast_2d(row_scale) / df
