#!/usr/bin/env python3
"""Counterpart of the reference's VehicleSimulation_Simulation.py on the HIP engine: lateral vehicle dynamics with the front and
rear tyre friction curves as two latent functions (src/Vehicle.py), Algorithm1 + Algorithm2, and the same .mat dictionary
(VehicleSimulation_Simulation.py:105-155; the reference's typo `online_T2_r = online_T2_f`, quirk Q10, is NOT reproduced).

    python examples/VehicleSimulation_Simulation.py [--particles 200] [--iterations 800] [--steps 1500] [--out plots/Vehicle.mat]
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from _marginal_driver import ROOT, posterior_mean, run_online_offline


def run(particles=200, iterations=800, steps=1500, seed=12345678, device=None, log=print):
    from pgas_amd import experiments

    pb = experiments.vehicle_marginal(T=steps, seed=seed)                  # src/Vehicle.py:14-292
    online, offline, times = run_online_offline(pb, particles, iterations, seed, device, log)
    on_X, on_mu, on_stats, on_w, _, _, on_Y, on_ll = online                # :24-40
    off_X, off_mu, off_w, off_stats, off_Y, off_ll = offline               # :69-83
    c = lambda a: a.cpu().numpy()  # noqa: E731
    bf, br = pb.basis
    alpha = lambda X, b: np.stack([b.alpha(X[t], pb.inputs[t]) for t in range(steps)])  # noqa: E731   (:41-43, :93-95: f_alpha over the traces)
    alpha_plot = np.linspace(-20 / 180 * np.pi, 20 / 180 * np.pi, 500)     # :100-103
    mu, B, C, E = 0.9, 10.0, 1.9, 0.97
    mu_true = mu * np.sin(C * np.arctan(B * (1 - E) * np.tan(alpha_plot) + E * np.arctan(B * np.tan(alpha_plot))))
    res = {"time": np.arange(steps) * 0.02, "alpha_plot": alpha_plot, "basis_plot": bf.map.batch(alpha_plot.reshape(-1, 1), None),
           "mu_true_plot": mu_true, "X": pb.X_true, "Y": pb.observations, "mu_f": pb.int_var_true[0], "mu_r": pb.int_var_true[1],
           "alpha_f": alpha(pb.X_true[:, None, :], bf)[:, 0], "alpha_r": alpha(pb.X_true[:, None, :], br)[:, 0], **times}
    for tag, X, MU, W, ST, Yp, LL in (("online", c(on_X), on_mu, on_w, on_stats, on_Y, on_ll), ("offline", c(off_X), off_mu, off_w, off_stats, off_Y, off_ll)):
        res.update({f"{tag}_Sigma_X": X, f"{tag}_Sigma_Y": c(Yp), f"{tag}_Sigma_mu_f": c(MU[0]), f"{tag}_Sigma_mu_r": c(MU[1]),
                    f"{tag}_Sigma_alpha_f": alpha(X, bf), f"{tag}_Sigma_alpha_r": alpha(X, br), f"{tag}_weights": c(W), f"{tag}_log_likelihood": c(LL)})
        for i, s in enumerate("fr"):
            for j in range(4):
                res[f"{tag}_T{j}_{s}"] = c(ST[i][j])
    for i, s in enumerate("fr"):
        for j in range(4):
            res[f"prior_T{j}_{s}"] = pb.GP_prior[i][j]
    return res


def friction_rmse(res, which="online", tyre="f"):
    """RMSE of the learned friction curve against the Pacejka truth on the slip angles the data visited."""
    prior = [res[f"prior_T{j}_{tyre}"] for j in range(4)]
    if which == "online":
        stats = [res[f"online_T{j}_{tyre}"][-1] for j in range(4)]
    else:
        stats = [np.mean(res[f"offline_T{j}_{tyre}"], axis=0) for j in range(4)]
    mean = posterior_mean(prior, stats)
    est = (res["basis_plot"] @ mean.T).reshape(-1)
    near = np.abs(res["alpha_plot"]) <= np.abs(res["alpha_" + tyre]).max()   # slip angles the simulated truth visited
    return float(np.sqrt(np.mean((est[near] - res["mu_true_plot"][near]) ** 2))), float(np.sqrt(np.mean(res["mu_true_plot"][near] ** 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=200)    # src/Vehicle.py:180
    ap.add_argument("--iterations", type=int, default=800)   # :181
    ap.add_argument("--steps", type=int, default=1500)       # t_end = 30 s at dt = 0.02 (:183-186)
    ap.add_argument("--out", default=os.path.join(ROOT, "plots", "Vehicle.mat"))
    args = ap.parse_args()
    res = run(args.particles, args.iterations, args.steps)
    import scipy.io

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    scipy.io.savemat(args.out, res)
    for which in ("online", "offline"):
        for tyre in "fr":
            r, s = friction_rmse(res, which, tyre)
            print(f"{which} mu_{tyre}: RMSE of the learned friction curve {r:.4f} (RMS of the true curve on the visited range: {s:.4f})")
    print("saved", args.out)


if __name__ == "__main__":
    main()
