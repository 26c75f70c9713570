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
* Vehicle-PGAS (config 3): src/Vehicle.py data (constants, Pacejka tyre curve, RK4, steering
  profile) with a plain-PGAS instantiation over a 3-D Hilbert basis of (yaw rate, lateral velocity,
  steering angle), M = 729 -- the "larger basis-function set".  The reference only runs Vehicle
  through Algorithm1/2 (marginalised); this PGAS instantiation, its observation
  model y = x + e and its basis scaling are the BUILD's interpretation of BASELINE.json configs[2].
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


# ---------------------------------------------------------------- Vehicle lateral dynamics (synthetic data)
def vehicle_pgas(T=2000, seed=12345678, M=729):
    m, I_zz, l_f, l_r, g, mu_x = 1720.0, 1827.5, 1.16, 1.47, 9.81, 0.9   # src/Vehicle.py:17-22
    mu, B, C, E = 0.9, 10.0, 1.9, 0.97                                    # :23-26
    dt = 0.02                                                             # :183
    t_end = T * dt
    time = np.arange(T) * dt
    F_zf, F_zr = m * g * l_r / (l_f + l_r), m * g * l_f / (l_f + l_r)     # :30-36

    def mu_y(alpha):                                                      # :40-47
        return mu * np.sin(C * np.arctan(B * (1 - E) * np.tan(alpha) + E * np.arctan(B * np.tan(alpha))))

    def f_alpha(x, u):                                                    # :51-58
        return u[0] - np.arctan((x[1] + x[0] * l_f) / u[1]), -np.arctan((x[1] - x[0] * l_r) / u[1])

    def dx(x, u, mf, mr):                                                 # :62-86
        dv_y = (F_zf * mf * np.cos(u[0]) + F_zr * mr + F_zf * mu_x * np.sin(u[0])) / m - u[1] * x[0]
        ddpsi = (l_f * F_zf * mf * np.cos(u[0]) - l_r * F_zr * mr + l_f * F_zf * mu_x * np.sin(u[0])) / I_zz
        return np.array([ddpsi, dv_y])

    ctrl = np.zeros((T, 2))                                               # :199-208
    ctrl[:, 0] = 10 / 180 * np.pi * np.sin(2 * np.pi * time / 5) * np.exp(-0.5 * (time - t_end / 2) ** 2 / (t_end / 5) ** 2)
    ctrl[:, 1] = 11.0
    x0, P0 = np.array([0.0, 0.0]), np.diag([1e-4, 1e-4])                  # :190-191
    R, Q = np.diag([0.001 / 180 * np.pi, 1e-3]), np.diag([1e-8, 1e-8])    # :195-196
    rng = np.random.default_rng(seed)
    X, Y = np.zeros((T, 2)), np.zeros((T, 2))
    X[0] = x0
    for i in range(1, T):                                                 # :226-257 (tyre forces frozen over the step, RK4 :90-100)
        s, u = X[i - 1], ctrl[i - 1]
        af, ar = f_alpha(s, u)
        mf, mr = mu_y(af), mu_y(ar)
        k1 = dx(s, u, mf, mr); k2 = dx(s + dt * k1 / 2, u, mf, mr); k3 = dx(s + dt * k2 / 2, u, mf, mr); k4 = dx(s + dt * k3, u, mf, mr)
        X[i] = s + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4) + np.sqrt(np.diag(Q)) * rng.standard_normal(2)
        Y[i] = X[i] + np.sqrt(np.diag(R)) * rng.standard_normal(2)       # build's observation model: both states, reference R
    basis, sd = generate_Hilbert_BasisFunction(M, np.array([[-1, 1], [-1, 1], [-1, 1]]), 0.5 / M, 20)  # as EMPS-729 (src/EMPS.py:101-107)
    prior = prior_mniw_2naturalPara(np.zeros((2, M)), np.diag(sd), np.eye(2), 2)
    bmap = basis.on([0, 1, 2], div=[1.0, 2.0, 0.25])      # |yaw rate| < 1 rad/s, |v_y| < 2 m/s, |steer| < 0.25 rad
    return Problem("Vehicle-PGAS", Y, ctrl, x0, P0, GaussianLikelihood(np.eye(2), R), bmap, prior, X)


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


# ================================================================ marginalised family (Algorithm1/2/3): problem definitions
@dataclass
class MarginalProblem:
    """One configuration of the reference's Algorithm1/Algorithm2 drivers.  `model(xp)` returns (transition_model, output_model)
    written against the array namespace xp (numpy or torch), batched over particles: the SAME arithmetic for the device mirror
    (pgas_amd.StateSpaceModel) and for the NumPy restatement used by the tests."""
    name: str
    observations: np.ndarray
    inputs: np.ndarray
    process_noise: np.ndarray
    output_noise: np.ndarray
    init_state_mean: np.ndarray
    init_state_cov: np.ndarray
    init_int_var_mean: list
    init_int_var_cov: list
    GP_prior: list
    basis: list                 # BasisMap per interface variable; basis[i].batch(state, input)
    forgetting_factor: float
    model: object
    X_true: np.ndarray
    int_var_true: list

    @property
    def T(self):
        return self.observations.shape[0]

    def ssm(self, cls, xp):
        f, g = self.model(xp)
        return cls(process_noise=self.process_noise, output_noise=self.output_noise, transition_model=f, output_model=g)

    def ssm_symbolic(self, cls):
        """The same model for pgas_amd.SymbolicStateSpaceModel: the factory itself, to be traced into one-launch programs."""
        return cls(process_noise=self.process_noise, output_noise=self.output_noise, model=self.model)

    def basis_fcn(self):
        """basis_fcn[i](state (N,n_x), input) -> (N,M); entries of `basis` are BasisMap descriptors or plain batched callables."""
        return [_BatchedBasis(b) if hasattr(b, "batch") else b for b in self.basis]


def smo_marginal(T=750, seed=12345678):
    """src/SingleMassOscillator.py:14-167: the spring-damper force F_sd is the latent function, interface variable of the RK4 model."""
    m, dt = 0.2, 0.02                                      # :17, :78
    pg = smo_pgas(T=T, seed=seed)                          # same data and basis (M = 41 on [-7.5, 7.5]^2)
    X = pg.X_true
    c1, c2, d1, d2 = 5.0, 2.0, 0.4, 0.4
    F_sd = c1 * X[:, 0] + c2 * X[:, 0] ** 3 + d1 * X[:, 1] / (1 + d2 * X[:, 1] * np.tanh(X[:, 1]))   # :24-29

    def model(xp):
        def dx(x, F, F_sd):                                # :32-33
            return xp.stack([x[:, 1], (-F_sd + F) / m], 1)

        def f_x(state, input, *int_var):                   # :36-44, :104-106
            F, Fs = input.reshape(-1)[0], int_var[0].reshape(-1)
            k1 = dx(state, F, Fs)
            k2 = dx(state + dt / 2.0 * k1, F, Fs)
            k3 = dx(state + dt / 2.0 * k2, F, Fs)
            k4 = dx(state + dt * k3, F, Fs)
            return state + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)

        def f_y(state, input, *int_var):                   # :47-48, :107
            return state[:, 0:1]

        return f_x, f_y

    basis = pg.basis_fcn.basis
    sd = np.diag(np.linalg.inv(pg.GP_prior[1]))            # eta1 = diag(sd)^-1
    prior = prior_mniw_2naturalPara(np.zeros((1, basis.M)), np.diag(sd), np.eye(1), 3)   # :63-68
    return MarginalProblem("SMO", pg.observations, pg.inputs.reshape(-1, 1), np.diag([5e-8, 5e-9]), np.array([[1e-3]]), np.array([0.0, 0.0]),
                           np.diag([1e-4, 1e-4]), [np.array([0.0])], [np.diag([1e-12])], [prior], [basis.on([0, 1])], 0.999, model, X, [F_sd])


def smo_two_component_marginal(T=750, seed=12345678):
    """NOT a configuration of the reference (all of them have scalar interface variables): the SingleMassOscillator with its force split
    into a spring part and a damper part, learnt as ONE latent function with n = 2 components over the same basis -- the case the
    reference's MNIW formulas (BI:18-108, eta0 (M, n), eta2 (n, n)) are written for.  Exercises the n > 1 path of Algorithm1."""
    m, dt = 0.2, 0.02
    pg = smo_pgas(T=T, seed=seed)
    X = pg.X_true
    c1, c2, d1, d2 = 5.0, 2.0, 0.4, 0.4
    F_s = c1 * X[:, 0] + c2 * X[:, 0] ** 3
    F_d = d1 * X[:, 1] / (1 + d2 * X[:, 1] * np.tanh(X[:, 1]))

    def model(xp):
        def dx(x, F, F_sd):
            return xp.stack([x[:, 1], (-F_sd + F) / m], 1)

        def f_x(state, input, *int_var):
            F, Fs = input.reshape(-1)[0], int_var[0][:, 0] + int_var[0][:, 1]
            k1 = dx(state, F, Fs)
            k2 = dx(state + dt / 2.0 * k1, F, Fs)
            k3 = dx(state + dt / 2.0 * k2, F, Fs)
            k4 = dx(state + dt * k3, F, Fs)
            return state + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)

        def f_y(state, input, *int_var):
            return state[:, 0:1]

        return f_x, f_y

    basis = pg.basis_fcn.basis
    sd = np.diag(np.linalg.inv(pg.GP_prior[1]))
    prior = prior_mniw_2naturalPara(np.zeros((2, basis.M)), np.diag(sd), np.array([[1.0, 0.2], [0.2, 0.5]]), 4)
    return MarginalProblem("SMO-2", pg.observations, pg.inputs.reshape(-1, 1), np.diag([5e-8, 5e-9]), np.array([[1e-3]]), np.array([0.0, 0.0]),
                           np.diag([1e-4, 1e-4]), [np.array([0.0, 0.0])], [np.array([[1e-12, 2e-13], [2e-13, 1e-12]])], [prior], [basis.on([0, 1])], 0.999,
                           model, X, [np.stack([F_s, F_d], 1)])


def toy_marginal(T=40, seed=12345678):
    """src/Toy_Example.py:14-128: no model knowledge, the transition IS the latent function (x_t = xi_{t-1})."""
    pt = toy(T=T, seed=seed)

    def model(xp):
        def f_x(state, input, *int_var):                   # :69
            return int_var[0].reshape(-1, 1)

        def f_y(state, input, *int_var):                   # :70 with f_y = identity (:22-23)
            return int_var[0].reshape(-1, 1)

        return f_x, f_y

    X = pt.X_true
    fx_true = 10 * np.sinc(X[:, 0] / 7)
    return MarginalProblem("Toy", pt.observations, pt.inputs, np.zeros((1, 1)), np.diag([4.0]), np.array([0.0]), np.diag([1e-4]),
                           [np.array([10 * np.sinc(0.0)])], [np.diag([4.0])], [pt.GP_prior], [pt.basis_fcn], 1.0, model, X, [fx_true])


class _BatchedBasis:
    """A descriptor as the batched callable Algorithm1/2/3 expect, keeping its one-call trajectory evaluation."""

    def __init__(self, b):
        self.b = b

    def __call__(self, state, input):
        return self.b.batch(state, input)

    def bind(self, ops):
        if hasattr(self.b, "bind"):
            self.b.bind(ops)

    def trajectory(self, states, inputs):
        return self.b.trajectory(states, inputs)


class _SlipAngleBasis:
    """basis_fcn_f / basis_fcn_r of src/Vehicle.py:146-153: the 1-D Hilbert basis evaluated at the front / rear tyre side-slip angle
    (f_alpha, :51-58).  Batched, NumPy or torch."""

    def __init__(self, basis, rear, l_f=1.16, l_r=1.47):
        self.map, self.rear, self.l_f, self.l_r = basis.on([0]), rear, l_f, l_r

    def alpha(self, state, input):
        xp = np if isinstance(state, np.ndarray) else __import__("torch")
        if self.rear:                                   # input (n_u,) or, along a trajectory, (T, n_u)
            return -xp.arctan((state[:, 1] - state[:, 0] * self.l_r) / input[..., 1])
        return input[..., 0] - xp.arctan((state[:, 1] + state[:, 0] * self.l_f) / input[..., 1])

    def batch(self, state, input):
        return self.map.batch(self.alpha(state, input).reshape(-1, 1), None)

    def bind(self, ops):
        self.map.bind(ops)

    def trajectory(self, states, inputs):
        return self.batch(states, inputs)

    def __call__(self, state, input):
        return self.batch(state, input)


def vehicle_marginal(T=1500, seed=12345678, M=20):
    """src/Vehicle.py:14-292: lateral vehicle dynamics, the two tyre friction curves mu_y(alpha_f), mu_y(alpha_r) are the latent
    functions (two scalar interface variables).  M = 20 is the reference's basis size; BASELINE.json configs[2] asks for a
    larger set (the device kernels take M <= 62)."""
    m, I_zz, l_f, l_r, g, mu_x = 1720.0, 1827.5, 1.16, 1.47, 9.81, 0.9    # :17-22
    dt = 0.02                                                              # :183
    pv = vehicle_pgas(T=T, seed=seed, M=27)                                # same simulated states and inputs (:199-257)
    X, ctrl = pv.X_true, pv.inputs
    F_zf, F_zr = m * g * l_r / (l_f + l_r), m * g * l_f / (l_f + l_r)     # :30-36
    mu, B, C, E = 0.9, 10.0, 1.9, 0.97

    def mu_y(alpha):                                                       # :40-47
        return mu * np.sin(C * np.arctan(B * (1 - E) * np.tan(alpha) + E * np.arctan(B * np.tan(alpha))))

    basis, sd = generate_Hilbert_BasisFunction(M, np.array([-30 / 180 * np.pi, 30 / 180 * np.pi]), 2 / 180 * np.pi, 50, idx_start=2, idx_step=2)   # :134-143
    bf, br = _SlipAngleBasis(basis, rear=False), _SlipAngleBasis(basis, rear=True)
    mu_f_true = mu_y(np.array([bf.alpha(X[t:t + 1], ctrl[t])[0] for t in range(T)]))
    mu_r_true = mu_y(np.array([br.alpha(X[t:t + 1], ctrl[t])[0] for t in range(T)]))

    def model(xp):
        def dxf(x, u, mf, mr):                                             # :62-86
            dv_y = (F_zf * mf * xp.cos(u[0]) + F_zr * mr + F_zf * mu_x * xp.sin(u[0])) / m - u[1] * x[:, 0]
            ddpsi = (l_f * F_zf * mf * xp.cos(u[0]) - l_r * F_zr * mr + l_f * F_zf * mu_x * xp.sin(u[0])) / I_zz
            return xp.stack([ddpsi, dv_y], 1)

        def f_x(state, input, *int_var):                                   # :90-100, :213-215
            mf, mr = int_var[0].reshape(-1), int_var[1].reshape(-1)
            k1 = dxf(state, input, mf, mr)
            k2 = dxf(state + dt * k1 / 2.0, input, mf, mr)
            k3 = dxf(state + dt * k2 / 2.0, input, mf, mr)
            k4 = dxf(state + dt * k3, input, mf, mr)
            return state + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)

        def f_y(state, input, *int_var):                                   # :104-131, :216-218
            mf, mr = int_var[0].reshape(-1), int_var[1].reshape(-1)
            dv_y = (F_zf * mf * xp.cos(input[0]) + F_zr * mr + F_zf * mu_x * xp.sin(input[0])) / m - input[1] * state[:, 0]
            return xp.tanh(xp.stack([state[:, 0], dv_y], 1))

        return f_x, f_y

    prior = prior_mniw_2naturalPara(np.zeros((1, M)), np.diag(sd), np.eye(1), 0)   # :157-173
    R, Q = np.diag([0.001 / 180 * np.pi, 1e-3]), np.diag([1e-8, 1e-8])    # :195-196
    # observations of THIS model: y = tanh([yaw rate, lateral acceleration]) + noise (:254-255)
    rng = np.random.default_rng(seed + 1)
    f, gy = model(np)
    Y = np.zeros((T, 2))
    for t in range(1, T):
        Y[t] = gy(X[t:t + 1], ctrl[t], np.array([[mu_f_true[t]]]), np.array([[mu_r_true[t]]]))[0] + np.sqrt(np.diag(R)) * rng.standard_normal(2)
    return MarginalProblem("Vehicle", Y, ctrl, Q, R, np.array([0.0, 0.0]), np.diag([1e-4, 1e-4]), [np.array([0.0]), np.array([0.0])],
                           [np.diag([1e-4]), np.diag([1e-4])], [prior, prior], [bf, br], 0.999, model, X, [mu_f_true, mu_r_true])


def emps_marginal(T=2000, seed=12345678):
    """src/EMPS.py:40-240: electro-mechanical positioning system, the friction force F(dq) is the latent function; synthetic data from
    the reference's linear friction model (:169-193), as in emps_pgas."""
    Mass, dt = 95.11, 0.01
    pe = emps_pgas(T=T, seed=seed, M=27)
    X, tau = pe.X_true, pe.inputs
    F_true = 203.5 * X[:, 1] + 20.39 * np.sign(X[:, 1]) - 3.16             # :169-173 rearranged: ddq = (tau - F) / M
    basis, sd = generate_Hilbert_BasisFunction(9, np.array([-0.2, 0.2]), 0.4 / 9, 20)   # :81-85
    prior = prior_mniw_2naturalPara(np.zeros((1, 9)), np.diag(sd), np.eye(1) * 4, 2)    # :92-96

    def model(xp):
        def dxf(x, t_, F):                                                 # :158-165
            return xp.stack([x[:, 1], (t_ - F) / Mass], 1)

        def f_x(state, input, *int_var):                                   # :177-183, :205-207
            t_, F = input.reshape(-1)[0], int_var[0].reshape(-1)
            k1 = dxf(state, t_, F)
            k2 = dxf(state + dt * k1 / 2, t_, F)
            k3 = dxf(state + dt * k2 / 2, t_, F)
            k4 = dxf(state + dt * k3, t_, F)
            return state + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)

        def f_y(state, input, *int_var):                                   # :197-198, :208
            return state[:, 0:1]

        return f_x, f_y

    return MarginalProblem("EMPS", pe.observations, tau.reshape(-1, 1), np.diag([1e-6, 1e-7]), np.diag([1e-4]), pe.init_state_mean, np.diag([1e-5, 1e-6]),
                           [np.array([0.0])], [np.diag([1e-12])], [prior], [basis.on([1])], 0.999, model, X, [F_true])
