#!/usr/bin/env python3
"""Counterpart of the reference's Toy_Example_Simulation.py on the HIP engine: Algorithm1 (online), Algorithm2 (offline) and plain
PGAS on the 40-step sinc example (src/Toy_Example.py), each followed by the posterior of the learned function on a plot grid
(:119-195).  The reference draws its figure; this script saves the same quantities to a .mat file.

    python examples/Toy_Example_Simulation.py [--iterations K] [--pgas-iterations K2] [--particles N] [--out plots/Toy_Example.mat]

Saved fields: online_* / offline_* / pgas_* (Sigma_X, log_likelihood, T0..T3, fcn_mean, fcn_var), x_plot, fx_true_plot, basis_plot,
prior_T0..T3, X, Y.  `run()` is the PGAS part alone (initial reference = the observations, f_y being the identity, :22-23);
`main()` takes the initial reference from an Algorithm1 run like the reference (:44-66).
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


def run_marginal(iterations=200, particles=200, seed=12345678, device=None, log=print):
    """Algorithm1 + Algorithm2 part (Toy_Example_Simulation.py:22-93, :119-165)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _marginal_driver import posterior_mean, run_online_offline

    import pgas_amd
    from pgas_amd import experiments

    pb = experiments.toy_marginal(seed=seed)
    online, offline, times = run_online_offline(pb, particles, iterations, seed, device, log)
    c = lambda a: a.cpu().numpy()  # noqa: E731
    x_plot = np.linspace(-30, 30, 500)
    basis_plot = pb.basis[0].batch(x_plot.reshape(-1, 1), None)
    res = {"online_Sigma_X": c(online[0]), "online_Sigma_xi": c(online[1][0]), "online_weights": c(online[3]), "online_log_likelihood": c(online[7]),
           "offline_Sigma_X": c(offline[0]), "offline_Sigma_xi": c(offline[1][0]), "offline_weights": c(offline[2]), "offline_log_likelihood": c(offline[5]),
           **times}
    for j in range(4):
        res[f"online_T{j}"], res[f"offline_T{j}"] = c(online[2][0][j]), c(offline[3][0][j])
    for tag, stats in (("online", [res[f"online_T{j}"][-1] for j in range(4)]), ("offline", [np.mean(res[f"offline_T{j}"], axis=0) for j in range(4)])):
        e = [np.asarray(pb.GP_prior[0][j]) + np.asarray(stats[j]).reshape(np.shape(pb.GP_prior[0][j])) for j in range(3)] + [pb.GP_prior[0][3] + float(np.reshape(stats[3], -1)[0])]
        std = pgas_amd.prior_mniw_2naturalPara_inv(*e)                      # :127-135, :146-154
        m, col, row, _ = pgas_amd.prior_mniw_Predictive(std[0], std[1], std[2], std[3], basis_plot)
        res[f"{tag}_fcn_mean"], res[f"{tag}_fcn_var"] = m, np.diag(col - 1) * row[0, 0]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=200, help="Algorithm2 iterations (src/Toy_Example.py:50)")
    ap.add_argument("--pgas-iterations", type=int, default=600)   # N_PGAS_iter * 3, src/Toy_Example.py:133
    ap.add_argument("--particles", type=int, default=200)    # :49
    ap.add_argument("--out", default=os.path.join(ROOT, "plots", "Toy_Example.mat"))
    ap.add_argument("--resample-before-propagate", action="store_true", help="corrected mode of the PGAS part (not the reference's behaviour)")
    args = ap.parse_args()
    res = run(args.pgas_iterations, args.particles, resample_before_propagate=args.resample_before_propagate)
    marg = run_marginal(args.iterations, args.particles)
    res.update(marg)
    import scipy.io

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    scipy.io.savemat(args.out, res)
    X = res["X"][:, 0]
    near = (res["x_plot"] > X.min()) & (res["x_plot"] < X.max())
    rmse = float(np.sqrt(np.mean((res["pgas_fcn_mean"][near] - res["fx_true_plot"][near]) ** 2)))
    print(f"saved {args.out}; RMSE of the posterior mean against 10 sinc(x/7) on the visited range [{X.min():.1f}, {X.max():.1f}]: PGAS {rmse:.3f}, "
          f"online {float(np.sqrt(np.mean((res['online_fcn_mean'][near] - res['fx_true_plot'][near]) ** 2))):.3f}, "
          f"offline {float(np.sqrt(np.mean((res['offline_fcn_mean'][near] - res['fx_true_plot'][near]) ** 2))):.3f}")


if __name__ == "__main__":
    main()
