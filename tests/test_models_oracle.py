"""The product's batched physical models (pgas_amd/experiments.py, NumPy and torch namespaces) against an independent per-particle
restatement of the reference's model functions (oracle/models_numpy.py; src/SingleMassOscillator.py:24-48, src/Vehicle.py:30-131,
src/EMPS.py:156-197): one transition and one output evaluation per random particle, the slip-angle basis inputs, and the
simulated data sets the experiments are built on."""
import numpy as np
import pytest
import torch

from common import experiments
from oracle import models_numpy as mo

RT = 1e-13


def _check_model(pb, ref, n_int, rng, scale_x, scale_u, scale_xi):
    X = rng.standard_normal((64, 2)) * scale_x
    U = np.atleast_1d(rng.standard_normal(np.size(pb.inputs[0])) * scale_u + (np.asarray(pb.inputs[1]) if np.size(pb.inputs[1]) else 0.0))
    XI = [rng.standard_normal((64, 1)) * scale_xi for _ in range(n_int)]
    for xp, conv in ((np, lambda a: a), (torch, lambda a: torch.as_tensor(a))):
        f, g = pb.model(xp)
        fx = np.asarray(f(conv(X), conv(U), *[conv(v) for v in XI]))
        gy = np.asarray(g(conv(X), conv(U), *[conv(v) for v in XI]))
        for p in range(64):
            args = [v[p, 0] for v in XI]
            u = U if U.size > 1 else U[0]
            np.testing.assert_allclose(fx[p], ref.transition_model(X[p], u, *args), rtol=RT, atol=1e-15, err_msg=f"{pb.name} transition ({xp.__name__})")
            np.testing.assert_allclose(gy[p].reshape(-1), np.atleast_1d(ref.output_model(X[p], u, *args)), rtol=RT, atol=1e-15,
                                       err_msg=f"{pb.name} output ({xp.__name__})")


def test_smo_model_and_data():
    rng = np.random.default_rng(1)
    pb = experiments.smo_marginal(T=60)
    _check_model(pb, mo.SMO, 1, rng, 2.0, 1.0, 5.0)
    # the simulated data follow the reference's loop (:122-130): F_sd frozen over the step, RK4 + process noise
    pg = experiments.smo_pgas(T=60)
    r2 = np.random.default_rng(12345678)
    Lq = np.linalg.cholesky(np.diag([5e-8, 5e-9]))
    x = np.zeros(2)
    for i in range(1, 60):
        F_sd = mo.SMO.F_spring(x[0]) + mo.SMO.F_damper(x[1])
        x = mo.SMO.f_x(x, pg.inputs[i - 1], F_sd, mo.SMO.dt) + Lq @ r2.standard_normal(2)
        r2.standard_normal()
        np.testing.assert_allclose(pg.X_true[i], x, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(pb.int_var_true[0], mo.SMO.F_spring(pb.X_true[:, 0]) + mo.SMO.F_damper(pb.X_true[:, 1]), rtol=1e-13)


def test_vehicle_model_slip_angles_and_data():
    rng = np.random.default_rng(2)
    pb = experiments.vehicle_marginal(T=50)
    _check_model(pb, mo.Vehicle, 2, rng, np.array([0.3, 0.8]), np.array([0.05, 0.0]), 0.5)
    # basis inputs: front / rear side-slip angles (:51-58, :146-153)
    X = rng.standard_normal((40, 2)) * np.array([0.3, 0.8])
    u = np.array([0.07, 11.0])
    bf, br = pb.basis
    af, ar = bf.alpha(X, u), br.alpha(X, u)
    for p in range(40):
        a_f, a_r = mo.Vehicle.f_alpha(X[p], u)
        assert abs(af[p] - a_f) < 1e-15 and abs(ar[p] - a_r) < 1e-15
    # simulated states: tyre forces frozen over the step (:240-247)
    pv = experiments.vehicle_pgas(T=50, M=27)
    r2 = np.random.default_rng(12345678)
    x = np.zeros(2)
    for i in range(1, 50):
        a_f, a_r = mo.Vehicle.f_alpha(x, pv.inputs[i - 1])
        x = mo.Vehicle.f_x(x, pv.inputs[i - 1], mo.Vehicle.mu_y(a_f), mo.Vehicle.mu_y(a_r), mo.Vehicle.dt) + np.sqrt([1e-8, 1e-8]) * r2.standard_normal(2)
        r2.standard_normal(2)
        np.testing.assert_allclose(pv.X_true[i], x, rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(pb.int_var_true[0][5], mo.Vehicle.mu_y(mo.Vehicle.f_alpha(pb.X_true[5], pb.inputs[5])[0]), rtol=1e-13)


def test_emps_model_and_data():
    rng = np.random.default_rng(3)
    pb = experiments.emps_marginal(T=80)
    _check_model(pb, mo.EMPS, 1, rng, 0.2, 30.0, 40.0)
    pe = experiments.emps_pgas(T=80, M=27)
    x = np.zeros(2)
    for i in range(1, 80):   # the reference's linear-friction model (:168-192) generates the synthetic data
        x = mo.EMPS.f_x_linModel(x, pe.inputs[i - 1], mo.EMPS.dt)
        np.testing.assert_allclose(pe.X_true[i], x, rtol=1e-12, atol=1e-16)
    # consistency of the two reference models: f_x with F = the linear friction force equals f_x_linModel while dq keeps its sign
    s = np.array([0.1, 0.05])
    F = 203.5 * s[1] + 20.39 * np.sign(s[1]) - 3.16
    k = mo.EMPS.dx(s, 12.0, F)
    np.testing.assert_allclose(k, mo.EMPS.dx_linModel(s, 12.0), rtol=1e-14)


def test_toy_model():
    pb = experiments.toy_marginal(T=20)
    f, g = pb.model(np)
    xi = np.linspace(-3, 3, 7).reshape(-1, 1)
    assert np.array_equal(f(np.zeros((7, 1)), None, xi), xi) and np.array_equal(g(np.zeros((7, 1)), None, xi), xi)
    np.testing.assert_allclose(pb.int_var_true[0], mo.Toy.f_x(pb.X_true[:, 0]), rtol=1e-14)
