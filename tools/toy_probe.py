"""Toy PGAS posterior quality probe (development aid): RMSE of the posterior mean vs 10 sinc(x/7)."""
import sys, os, importlib.util
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("toy_driver", os.path.join(ROOT, "examples", "Toy_Example_Simulation.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import pgas_amd
from pgas_amd import experiments
pb = experiments.toy()
basis = pb.basis_fcn.basis
X = pb.X_true
Phi = np.stack([basis(X[t]) for t in range(pb.T - 1)]); Xp = X[1:]
std = pgas_amd.prior_mniw_2naturalPara_inv(pb.GP_prior[0] + Phi.T @ Xp, pb.GP_prior[1] + Phi.T @ Phi, pb.GP_prior[2] + Xp.T @ Xp, pb.GP_prior[3] + pb.T - 1)
x_plot = np.linspace(-30, 30, 500)
bp = np.stack([basis(np.array([x])) for x in x_plot])
fm = pgas_amd.prior_mniw_Predictive(std[0], std[1], std[2], std[3], bp)[0]
ft = 10 * np.sinc(x_plot / 7)
lo, hi = np.percentile(X[:, 0], 10), np.percentile(X[:, 0], 90)
near = (x_plot > lo) & (x_plot < hi)
print("range", lo, hi, "RMSE given the TRUE states:", np.sqrt(np.mean((fm[near] - ft[near]) ** 2)))
for corrected in (False, True):
    for K, N in ((300, 200), (1200, 200), (300, 4000)):
        res = m.run(iterations=K, particles=N, resample_before_propagate=corrected)
        err = res["pgas_fcn_mean"][near] - res["fx_true_plot"][near]
        burn = K // 3
        print("corrected" if corrected else "reference", "K", K, "N", N, "RMSE", np.sqrt(np.mean(err ** 2)),
              "traj rmse vs truth", np.sqrt(np.mean((res["pgas_Sigma_X"][:, burn:, 0] - X) ** 2)), flush=True)
