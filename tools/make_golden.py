"""Writes tests/golden/*.json.

There are no reference outputs to take vectors from (the reference's plots/*.mat are Git-LFS
stubs and JAX is not installable here).  The fixtures are therefore
 (1) analytic known answers from SURVEY.md section 8c (basis index tables obtained there by tracing
     reference src/BasisFunctions.py:24-57 by hand, spectral-density ranges), written down here as data, and
 (2) regression vectors of the canonical C oracle (oracle/pgas_canon.c) on tiny problems, so that
     an accidental change of the canonical arithmetic is caught on CPU.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(GOLD, exist_ok=True)
    tables = {
        "smo_2d_41": {"num_fcn": 41, "domain": [[-7.5, 7.5], [-7.5, 7.5]], "lengthscale": 15 / 41, "scale": 100,
                      "first": [[1, 1], [1, 2], [2, 1], [2, 2], [1, 3], [3, 1]], "max": [7, 7], "sd_range": [70.3, 83.6]},
        "emps_3d_729": {"num_fcn": 729, "domain": [[-1, 1], [-1, 1], [-1, 1]], "lengthscale": 0.5 / 729, "scale": 20,
                        "first": [[1, 1, 1], [1, 1, 2], [1, 2, 1], [2, 1, 1], [1, 2, 2], [2, 1, 2]], "max": [11, 11, 11],
                        "sd_range": [1.016e-7, 1.016e-7]},
        "vehicle_1d_20_even": {"num_fcn": 20, "domain": [[-0.5235987755982988, 0.5235987755982988]], "idx_start": 2, "idx_step": 2,
                               "first": [[2], [4], [6]], "max": [40]},
        "toy_1d_40": {"num_fcn": 40, "domain": [[-30, 30]], "lengthscale": 3, "scale": 50, "first": [[1], [2], [3]], "max": [40],
                      "sd_range": [1.006e-6, 371.4]},
        "emps_1d_9": {"num_fcn": 9, "domain": [[-0.2, 0.2]], "lengthscale": 0.4 / 9, "scale": 20, "first": [[1], [2]], "max": [9],
                      "sd_range": [1.6e-2, 2.10]},
    }
    json.dump(tables, open(os.path.join(GOLD, "basis_index_tables.json"), "w"), indent=1)

    from common import canon_model, experiments

    out = {}
    for name, pb, N in (("smo", experiments.smo_pgas(T=10), 130), ("toy", experiments.toy(T=12), 70), ("emps27", experiments.emps_pgas(T=8, M=27), 1100)):
        A, S = experiments.initial_params(pb)
        cm = canon_model(pb, N)
        LS, LSinv, cS = cm.chol_parts(S)
        traj, X, ANC, lw = cm.sweep(12345678, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov))
        out[name] = {
            "N": N, "T": pb.T, "seed": 12345678,
            "traj_hex": [float(v).hex() for v in traj.reshape(-1)],
            "anc_last": ANC[-1].tolist(),
            "anc_sum": [int(r.sum()) for r in ANC],
            "logw_last_hex": [float(v).hex() for v in lw[:8]],
        }
    json.dump(out, open(os.path.join(GOLD, "canon_sweeps.json"), "w"), indent=1)

    # (3) regression vectors of the NumPy restatement of the marginalised family (oracle/marginal_numpy.py) on the canonical
    #     Philox streams: tiny Algorithm1 / Algorithm3 runs, so that a change of the restatement, of the problem definitions or of
    #     the random-number streams is caught on CPU, and the GPU tests have a committed target besides the live oracle.
    from common import CanonRand, marginal_oracle
    from oracle import marginal_numpy as mo

    marg = {}
    for name, mk, N in (("smo", experiments.smo_marginal, 48), ("toy", experiments.toy_marginal, 40), ("vehicle", experiments.vehicle_marginal, 36)):
        pb = mk(T=6)
        o1 = marginal_oracle(pb, N)(CanonRand(12345678, N))
        a3 = marginal_oracle(pb, N, "Algorithm3")
        ref_stats = mo.trajectory_stats(a3, pb.X_true, list(pb.int_var_true))
        traj, ivt, tr = a3(CanonRand(12345678, N), pb.X_true, list(pb.int_var_true), ref_stats)
        marg[name] = {
            "N": N, "T": pb.T, "seed": 12345678,
            "alg1_ancestors": o1[4].tolist(),
            "alg1_state_last": o1[0][-1].reshape(-1).tolist(),
            "alg1_int_var_last": [v[-1].reshape(-1).tolist() for v in o1[1]],
            "alg1_weights_last": o1[3][-1].tolist(),
            "alg1_T1_trace_diag_last": [np.diag(s[1][-1]).tolist() for s in o1[2]],
            "alg3_ancestors": tr["ancestor_trace"].tolist(), "alg3_idx": int(tr["idx"]),
            "alg3_state_traj": np.asarray(traj).reshape(-1).tolist(),
            "alg3_int_var_traj": [np.asarray(v).reshape(-1).tolist() for v in ivt],
        }
    json.dump(marg, open(os.path.join(GOLD, "marginal_runs.json"), "w"), indent=1)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
