#!/usr/bin/env python3
"""PGAS part of the reference's Toy_Example_Simulation.py (:96-107, :119-195) on the HIP engine.

The reference driver also runs Algorithm1/Algorithm2 (the marginalised family, SURVEY 8 f1 -- not built yet) and takes the
initial reference trajectory from an Algorithm1 run (:44-66).  Here the initial reference is the observation sequence itself
(f_y is the identity, src/Toy_Example.py:22-23), which the Gibbs sampler forgets after a few iterations.

    python examples/Toy_Example_Simulation.py [--iterations K] [--particles N] [--out plots/Toy_Example_PGAS.mat]

Saved fields follow the reference's names where it has them: pgas_Sigma_X (T,K,1), pgas_log_likelihood (T,K), pgas_T0..T3,
x_plot, fx_true_plot, basis_plot, pgas_fcn_mean, pgas_fcn_var, prior_T0..T3, X, Y.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(iterations=600, particles=200, seed=12345678, device=None, resample_before_propagate=False):
    import pgas_amd
    from pgas_amd import experiments

    pb = experiments.toy(seed=seed)                                        # src/Toy_Example.py:17-96
    pg = pgas_amd.PGAS(particles, iterations, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn,
                       pb.GP_prior, pb.basis_fcn, device=device, resample_before_propagate=resample_before_propagate)   # :131-147
    key = pgas_amd.random.key(seed)
    Sigma_X, loglik = pg(key, pb.observations.reshape(-1, 1))             # Toy_Example_Simulation.py:102-105
    Sigma_X, loglik = Sigma_X.cpu().numpy(), loglik.cpu().numpy()          # (T,K,1), (T,K)
    basis = pb.basis_fcn.basis
    # sufficient statistics of every sampled trajectory, averaged over the K samples (:108-114)
    K = Sigma_X.shape[1]
    T0 = np.zeros_like(pb.GP_prior[0]); T1 = np.zeros_like(pb.GP_prior[1]); T2 = np.zeros_like(pb.GP_prior[2]); T3 = 0.0
    for k in range(K):
        Phi = np.stack([basis(Sigma_X[t, k]) for t in range(pb.T - 1)])   # (T-1, M)
        Xp = Sigma_X[1:, k]
        T0 += Phi.T @ Xp; T1 += Phi.T @ Phi; T2 += Xp.T @ Xp; T3 += pb.T - 1
    T0 /= K; T1 /= K; T2 /= K; T3 /= K
    x_plot = np.linspace(-30, 30, 500)                                     # :122-124
    fx_true_plot = 10 * np.sinc(x_plot / 7)
    basis_plot = np.stack([basis(np.array([x])) for x in x_plot])
    std = pgas_amd.prior_mniw_2naturalPara_inv(pb.GP_prior[0] + T0, pb.GP_prior[1] + T1, pb.GP_prior[2] + T2, pb.GP_prior[3] + T3)  # :166-173
    fcn_mean, col_scale, row_scale, _ = pgas_amd.prior_mniw_Predictive(std[0], std[1], std[2], std[3], basis_plot)                  # :174-180
    fcn_var = np.diag(col_scale - 1) * row_scale[0, 0]                     # :181
    return {
        "pgas_Sigma_X": Sigma_X, "pgas_log_likelihood": loglik, "pgas_T0": T0, "pgas_T1": T1, "pgas_T2": T2, "pgas_T3": T3,
        "x_plot": x_plot, "fx_true_plot": fx_true_plot, "basis_plot": basis_plot, "pgas_fcn_mean": fcn_mean, "pgas_fcn_var": fcn_var,
        "prior_T0": pb.GP_prior[0], "prior_T1": pb.GP_prior[1], "prior_T2": pb.GP_prior[2], "prior_T3": pb.GP_prior[3],
        "X": pb.X_true, "Y": pb.observations,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=600)   # N_PGAS_iter * 3, src/Toy_Example.py:133
    ap.add_argument("--particles", type=int, default=200)    # :49
    ap.add_argument("--out", default=os.path.join(ROOT, "plots", "Toy_Example_PGAS.mat"))
    ap.add_argument("--resample-before-propagate", action="store_true", help="corrected mode (not the reference's behaviour)")
    args = ap.parse_args()
    res = run(args.iterations, args.particles, resample_before_propagate=args.resample_before_propagate)
    import scipy.io

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    scipy.io.savemat(args.out, res)
    X = res["X"][:, 0]
    near = (res["x_plot"] > X.min()) & (res["x_plot"] < X.max())
    rmse = float(np.sqrt(np.mean((res["pgas_fcn_mean"][near] - res["fx_true_plot"][near]) ** 2)))
    print(f"saved {args.out}; RMSE of the posterior mean against 10 sinc(x/7) on the visited range [{X.min():.1f}, {X.max():.1f}]: {rmse:.3f}")


if __name__ == "__main__":
    main()
