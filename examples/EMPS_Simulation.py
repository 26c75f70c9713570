#!/usr/bin/env python3
"""PGAS part of the reference's EMPS_Simulation.py (:95-118, :128-161) on the HIP engine, with synthetic data.

DATA_EMPS.mat is not distributed with the reference; the data come from its own linear-friction model (src/EMPS.py:169-193,
see pgas_amd/experiments.py::emps_pgas).  The reference driver also runs Algorithm1/2 (not built) and takes the
initial reference trajectory from Algorithm1; here it is (measured position, finite-difference velocity).

    python examples/EMPS_Simulation.py [--iterations K] [--particles N] [--steps T] [--out plots/EMPS_PGAS.mat]

Saved fields (the reference's names, EMPS_Simulation.py:128-160): offline_Sigma_X_PGAS (T,K,2), offline_log_likelihood_PGAS (T,K),
time, Y, X, prior_T0..T3 (the PGAS prior), plus PGAS_T0..T3 (posterior statistics) and PGAS_mean (2,M).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
        "prior_T0": pb.GP_prior[0], "prior_T1": pb.GP_prior[1], "prior_T2": pb.GP_prior[2], "prior_T3": pb.GP_prior[3],
        "PGAS_T0": post[0], "PGAS_T1": post[1], "PGAS_T2": post[2], "PGAS_T3": post[3], "PGAS_mean": mean,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=30)
    ap.add_argument("--particles", type=int, default=200)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--out", default=os.path.join(ROOT, "plots", "EMPS_PGAS.mat"))
    ap.add_argument("--resample-before-propagate", action="store_true", help="corrected mode (not the reference's behaviour)")
    args = ap.parse_args()
    res = run(args.iterations, args.particles, args.steps, resample_before_propagate=args.resample_before_propagate)
    import scipy.io

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    scipy.io.savemat(args.out, res)
    pos_rmse = float(np.sqrt(np.mean((res["offline_Sigma_X_PGAS"][:, -1, 0] - res["X"][:, 0]) ** 2)))
    print(f"saved {args.out}; position RMSE of the last sampled trajectory against the simulated truth: {pos_rmse:.4f}")


if __name__ == "__main__":
    main()
