#!/usr/bin/env python3
"""Counterpart of the reference's EMPS_Simulation.py on the HIP engine, with synthetic data: Algorithm1 (online), Algorithm2 (offline,
friction force F(dq) as the latent function, src/EMPS.py:76-98,200-240) and the plain-PGAS baseline over the 729-function basis
(:99-123,243-255).

DATA_EMPS.mat / DATA_EMPS_PULSES.mat are not distributed with the reference; the data come from its own linear-friction model
(src/EMPS.py:169-193, see pgas_amd/experiments.py), and the validation RMSEs (:128-157) are computed on a second synthetic
input sequence instead of the pulse measurements.

    python examples/EMPS_Simulation.py [--iterations K] [--pgas-iterations K2] [--particles N] [--steps T] [--out plots/EMPS.mat]

Saved fields follow EMPS_Simulation.py:128-160: online_* / offline_* (Sigma_X, Sigma_F, Sigma_Y, weights, log_likelihood, T0..T3),
offline_Sigma_X_PGAS, offline_log_likelihood_PGAS, time, dq_plot, basis_plot, prior_T0..T3, RMSE_Alg2, RMSE_PGAS, Y, X
(plus prior_T0..3_PGAS, PGAS_T0..T3 and PGAS_mean for the baseline's own statistics).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_marginal(iterations=30, particles=200, steps=2000, seed=12345678, device=None, log=print):
    """Algorithm1 + Algorithm2 part (EMPS_Simulation.py:26-93)."""
    from _marginal_driver import posterior_mean, run_online_offline
    from pgas_amd import experiments

    pb = experiments.emps_marginal(T=steps, seed=seed)
    online, offline, times = run_online_offline(pb, particles, iterations, seed, device, log)
    on_X, on_F, on_stats, on_w, _, _, on_Y, on_ll = online
    off_X, off_F, off_w, off_stats, off_Y, off_ll = offline
    c = lambda a: a.cpu().numpy()  # noqa: E731
    dq_plot = np.linspace(-0.15, 0.15, 500)                                 # :122-123
    res = {"online_Sigma_X": c(on_X), "online_Sigma_F": c(on_F[0]), "online_Sigma_Y": c(on_Y), "online_weights": c(on_w), "online_log_likelihood": c(on_ll),
           "offline_Sigma_X": c(off_X), "offline_Sigma_F": c(off_F[0]), "offline_Sigma_Y": c(off_Y), "offline_weights": c(off_w),
           "offline_log_likelihood": c(off_ll), "dq_plot": dq_plot, "basis_plot": pb.basis[0].basis.on([0]).batch(dq_plot.reshape(-1, 1), None), **times}
    for j in range(4):
        res[f"online_T{j}"], res[f"offline_T{j}"], res[f"prior_T{j}"] = c(on_stats[0][j]), c(off_stats[0][j]), pb.GP_prior[0][j]
    res["offline_mean"] = posterior_mean(pb.GP_prior[0], [np.mean(res[f"offline_T{j}"], axis=0) for j in range(4)])   # :84-89
    return res, pb


def validation_rmse(offline_mean, pgas_mean, marg_pb, pgas_pb, steps=600):
    """EMPS_Validation_Simulation (src/EMPS.py:128-157) on a synthetic pulse input: open-loop simulation with the two learned models
    against the reference's linear-friction truth."""
    dt, Mass = 0.01, 95.11
    tt = np.arange(steps) * dt
    tau = 45.0 * np.sign(np.sin(2 * np.pi * tt / 1.5))   # keeps the velocity inside the basis domain [-0.2, 0.2] (src/EMPS.py:82-84)
    f_np, _ = marg_pb.model(np)

    def truth(s, u):
        return np.array([s[1], (u - 203.5 * s[1] - 20.39 * np.sign(s[1]) + 3.16) / Mass])

    X = np.zeros((steps, 2)); Xa = np.zeros((steps, 2)); Xp = np.zeros((steps, 2))
    for i in range(1, steps):
        s, u = X[i - 1], tau[i - 1]
        k1 = truth(s, u); k2 = truth(s + dt * k1 / 2, u); k3 = truth(s + dt * k2 / 2, u); k4 = truth(s + dt * k3, u)
        X[i] = s + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        F = (offline_mean @ marg_pb.basis[0].batch(Xa[i - 1:i], None)[0])[0]
        Xa[i] = f_np(Xa[i - 1:i], np.array([u]), np.array([[F]]))[0]
        Xp[i] = pgas_mean @ pgas_pb.basis_fcn(Xp[i - 1], np.array([u]))
    return float(np.sqrt(np.mean((Xa[:, 0] - X[:, 0]) ** 2))), float(np.sqrt(np.mean((Xp[:, 0] - X[:, 0]) ** 2)))


def run(iterations=30, particles=200, steps=2000, seed=12345678, device=None, resample_before_propagate=False):
    import torch

    import pgas_amd
    from pgas_amd import experiments

    pb = experiments.emps_pgas(T=steps, seed=seed)
    pg = pgas_amd.PGAS(particles, iterations, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn,
                       pb.GP_prior, pb.basis_fcn, device=device, resample_before_propagate=resample_before_propagate)  # src/EMPS.py:243-255
    dt = 0.01
    y = pb.observations
    init_ref = np.stack([y, np.gradient(y, dt)], axis=1)
    Sigma_X, loglik = pg(pgas_amd.random.key(seed), init_ref)              # EMPS_Simulation.py:98-101
    eng = pg.cSMC.engine
    K = Sigma_X.shape[1]
    acc = None
    for k in range(K):                                                      # :104-116, the sums on the device (MFMA SYRK)
        st = eng.suffstats(Sigma_X[:, k].contiguous())
        acc = [a + b for a, b in zip(acc, st)] if acc else list(st)
    T0, T1, T2, T3 = [a / K for a in acc]
    post = (pb.GP_prior[0] + T0.cpu().numpy(), pb.GP_prior[1] + T1.cpu().numpy(), pb.GP_prior[2] + T2.cpu().numpy(), pb.GP_prior[3] + T3)
    mean = pgas_amd.prior_mniw_2naturalPara_inv(*post)[0]                   # :117
    torch.cuda.synchronize()
    return {
        "offline_Sigma_X_PGAS": Sigma_X.cpu().numpy(), "offline_log_likelihood_PGAS": loglik.cpu().numpy(),
        "time": np.arange(steps) * dt, "Y": pb.observations, "X": pb.X_true,
        "prior_T0_PGAS": pb.GP_prior[0], "prior_T1_PGAS": pb.GP_prior[1], "prior_T2_PGAS": pb.GP_prior[2], "prior_T3_PGAS": pb.GP_prior[3],
        "PGAS_T0": post[0], "PGAS_T1": post[1], "PGAS_T2": post[2], "PGAS_T3": post[3], "PGAS_mean": mean, "_pb": pb,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=30, help="Algorithm2 iterations (reference: 800)")
    ap.add_argument("--pgas-iterations", type=int, default=30, help="plain-PGAS iterations (reference: 2400)")
    ap.add_argument("--particles", type=int, default=200)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--out", default=os.path.join(ROOT, "plots", "EMPS.mat"))
    ap.add_argument("--resample-before-propagate", action="store_true", help="corrected mode of the PGAS baseline (not the reference's behaviour)")
    args = ap.parse_args()
    marg, mpb = run_marginal(args.iterations, args.particles, args.steps)
    res = run(args.pgas_iterations, args.particles, args.steps, resample_before_propagate=args.resample_before_propagate)
    ppb = res.pop("_pb")
    res.update(marg)
    res["RMSE_Alg2"], res["RMSE_PGAS"] = validation_rmse(marg["offline_mean"], res["PGAS_mean"], mpb, ppb)
    print(f"RMSE_Alg2: {res['RMSE_Alg2']:.5f}\nRMSE_PGAS: {res['RMSE_PGAS']:.5f}")
    import scipy.io

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    scipy.io.savemat(args.out, res)
    pos_rmse = float(np.sqrt(np.mean((res["offline_Sigma_X_PGAS"][:, -1, 0] - res["X"][:, 0]) ** 2)))
    print(f"saved {args.out}; position RMSE of the last sampled trajectory against the simulated truth: {pos_rmse:.4f}")


if __name__ == "__main__":
    main()
