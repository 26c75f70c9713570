#!/usr/bin/env python3
"""Counterpart of the reference's SingleMassOscillator_Simulation.py (BASELINE.json configs[0]) on the HIP engine:
Algorithm1 (online), a second Algorithm1 run for the initial reference (quirk Q13 reproduced: the index is drawn from the
cumulative sum of the FLATTENED (T,N) weights, :55), Algorithm2 (offline Particle Gibbs), and the same .mat dictionary (:94-125).

    python examples/SingleMassOscillator_Simulation.py [--particles 200] [--iterations 800] [--steps 750] [--out plots/SingleMassOscillator.mat]

Data: the reference simulates with jax.random; here NumPy's generator with the same seed value (pgas_amd/experiments.py).
"""
from __future__ import annotations

import argparse
import os
import sys
import time as _time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(particles=200, iterations=800, steps=750, seed=12345678, device=None, log=print):
    import torch

    import pgas_amd
    from pgas_amd import experiments
    from pgas_amd import random as prng

    pb = experiments.smo_marginal(T=steps, seed=seed)                     # src/SingleMassOscillator.py:14-139
    ssm = pb.ssm_symbolic(pgas_amd.SymbolicStateSpaceModel)   # the model callables traced into one-launch programs (StateSpaceModel + torch callables works the same)
    common = dict(observations=pb.observations, inputs=pb.inputs, SSM=ssm, init_state_mean=pb.init_state_mean, init_state_cov=pb.init_state_cov,
                  init_int_var_mean=pb.init_int_var_mean, init_int_var_cov=pb.init_int_var_cov, GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn(),
                  device=device)
    alg1 = pgas_amd.Algorithm1(N_samples=particles, forgetting_factor=pb.forgetting_factor, **common)       # :142-154
    alg2 = pgas_amd.Algorithm2(N_samples=particles, N_iterations=iterations, **common)                      # :157-169
    key = prng.key(seed)
    key, key_sim = prng.split(key, 2)
    t0 = _time.perf_counter()
    on_X, on_F, on_stats, on_w, _, _, on_Y, on_ll = alg1(key_sim)                                            # :21-31
    torch.cuda.synchronize()
    t_online = _time.perf_counter() - t0
    log(f"online (Algorithm1): N={particles}, T={steps}: {t_online:.2f} s = {particles * (steps - 1) / t_online:.3e} particle-steps/s")
    key, key_sim, key_traj = prng.split(key, 3)
    r_X, r_F, _, r_w, r_anc, _, _, _ = alg1(key_sim)                                                        # :43-53
    u = float(prng.uniform(key_traj, 1)[0])
    idx = int(np.searchsorted(np.cumsum(r_w.cpu().numpy()), u))                                              # :54 (Q13: flattened weights)
    idx = min(idx, particles - 1)
    init_ref_state = pgas_amd.reconstruct_trajectory(r_X, r_anc, idx)                                       # :55
    init_ref_int_var = [pgas_amd.reconstruct_trajectory(r_F[i], r_anc, idx) for i in range(len(r_F))]       # :56-60
    t0 = _time.perf_counter()
    off_X, off_F, off_w, off_stats, off_Y, off_ll = alg2(key, init_ref_state, init_ref_int_var)             # :63-71
    torch.cuda.synchronize()
    t_offline = _time.perf_counter() - t0
    log(f"offline (Algorithm2): K={iterations}: {t_offline:.2f} s = {particles * (steps - 1) * max(iterations - 1, 1) / t_offline:.3e} particle-steps/s")
    c = lambda a: a.cpu().numpy()  # noqa: E731
    x_plt = np.linspace(-3.5, 3.5, 50)                                                                       # :80-86
    gx, gy = np.meshgrid(x_plt, x_plt, indexing="xy")
    X_plot = np.vstack([gx.flatten(), gy.flatten()]).T
    basis_plot = pb.basis[0].batch(X_plot, None)
    F_true = 5.0 * X_plot[:, 0] + 2.0 * X_plot[:, 0] ** 3 + 0.4 * X_plot[:, 1] / (1 + 0.4 * X_plot[:, 1] * np.tanh(X_plot[:, 1]))   # :89-91
    return {
        "offline_Sigma_X": c(off_X), "offline_Sigma_Y": c(off_Y), "offline_Sigma_F": c(off_F[0]), "offline_weights": c(off_w),
        "offline_log_likelihood": c(off_ll), "offline_T0": c(off_stats[0][0]), "offline_T1": c(off_stats[0][1]), "offline_T2": c(off_stats[0][2]),
        "offline_T3": c(off_stats[0][3]),
        "online_Sigma_X": c(on_X), "online_Sigma_Y": c(on_Y), "online_Sigma_F": c(on_F[0]), "online_weights": c(on_w), "online_log_likelihood": c(on_ll),
        "online_T0": c(on_stats[0][0]), "online_T1": c(on_stats[0][1]), "online_T2": c(on_stats[0][2]), "online_T3": c(on_stats[0][3]),
        "time": np.arange(steps) * 0.02, "X_plot": X_plot, "basis_plot": basis_plot, "F_sd_true_plot": F_true,
        "prior_T0": pb.GP_prior[0][0], "prior_T1": pb.GP_prior[0][1], "prior_T2": pb.GP_prior[0][2], "prior_T3": pb.GP_prior[0][3],
        "X": pb.X_true, "Y": pb.observations, "F_sd": pb.int_var_true[0],
        "seconds_online": t_online, "seconds_offline": t_offline,
    }


def posterior_force_rmse(res, which="online"):
    """RMSE of the learned spring-damper force against the truth on the part of the plot grid the trajectory visited."""
    import pgas_amd

    if which == "online":
        stats = [res["prior_T%d" % j] + res["online_T%d" % j][-1] for j in range(4)]
    else:
        stats = [res["prior_T%d" % j] + np.mean(res["offline_T%d" % j], axis=0) for j in range(4)]
    mean = pgas_amd.prior_mniw_2naturalPara_inv(stats[0].reshape(-1, 1), stats[1], np.reshape(stats[2], (1, 1)), float(np.reshape(stats[3], -1)[0]))[0]
    F = (res["basis_plot"] @ mean.T).reshape(-1)
    X = res["X"]
    near = (np.abs(res["X_plot"][:, 0]) < np.abs(X[:, 0]).max()) & (np.abs(res["X_plot"][:, 1]) < np.abs(X[:, 1]).max())
    return float(np.sqrt(np.mean((F[near] - res["F_sd_true_plot"][near]) ** 2))), float(np.sqrt(np.mean(res["F_sd_true_plot"][near] ** 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=200)    # src/SingleMassOscillator.py:74
    ap.add_argument("--iterations", type=int, default=800)   # :75
    ap.add_argument("--steps", type=int, default=750)        # t_end = 15 s at dt = 0.02 (:76-80)
    ap.add_argument("--out", default=os.path.join(ROOT, "plots", "SingleMassOscillator.mat"))
    args = ap.parse_args()
    res = run(args.particles, args.iterations, args.steps)
    import scipy.io

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    scipy.io.savemat(args.out, res)
    for which in ("online", "offline"):
        r, s = posterior_force_rmse(res, which)
        print(f"{which}: RMSE of the learned F_sd on the visited region {r:.3f} (RMS of the true force there: {s:.3f})")
    print("saved", args.out)


if __name__ == "__main__":
    main()
