"""Pins the NumPy restatement of the marginalised family (oracle/marginal_numpy.py: reference src/Algorithm1.py, Algorithm3.py,
StateSpaceModel.py) by analytic properties -- the JAX reference cannot run here (parity unpinned)."""
import numpy as np
import pytest

from common import CanonRand, canon, experiments, marginal_oracle
from oracle import marginal_numpy as mo
from oracle import pgas_numpy as o

SEED = 12345678


def test_student_t_and_gamma_samplers_are_correctly_distributed():
    from scipy import stats

    for a in (0.5, 1.0, 7.25, 300.0):
        g = canon.gamma(SEED, 40, 3, 0, np.full(40000, a))
        assert stats.kstest(g, "gamma", args=(a,)).pvalue > 1e-3
    for nu in (1.0, 4.0, 751.0):
        t = canon.student_t(SEED, 33, 5, 0, np.full(40000, nu))
        assert stats.kstest(t, "t", args=(nu,)).pvalue > 1e-3
    # addressing: a particle's draw does not depend on how many particles are generated with it
    assert np.array_equal(canon.student_t(SEED, 33, 5, 10, np.full(5, 4.0)), canon.student_t(SEED, 33, 5, 0, np.full(15, 4.0))[10:])


def test_batched_mniw_algebra_matches_per_particle_functions():
    rng = np.random.default_rng(2)
    N, M = 6, 7
    B = rng.standard_normal((N, M, M)); eta1 = B @ np.swapaxes(B, 1, 2) + M * np.eye(M)
    eta0 = rng.standard_normal((N, M, 1)); eta2 = np.einsum("nmk,nml,nlj->nkj", eta0, np.linalg.inv(eta1), eta0) + 2.0
    mean, col_cov, row_scale, _ = mo._natural_inv_b(eta0, eta1, eta2, 5.0)
    for p in range(N):
        m1, c1, r1, _ = o.prior_mniw_2naturalPara_inv(eta0[p], eta1[p], eta2[p], 5.0)
        assert np.allclose(mean[p], m1) and np.allclose(col_cov[p], c1) and np.allclose(row_scale[p], r1)
        assert np.allclose(mo._mniw_mean_b(eta0, eta1)[p], o.prior_mniw_mean(eta0[p], eta1[p]))
        assert np.isclose(mo._log_base_measure_b(eta0, eta1, eta2, np.full(N, 5.0))[p], o.prior_mniw_log_base_measure(eta0[p], eta1[p], eta2[p], 5.0))


@pytest.mark.parametrize("name", ["smo", "toy", "vehicle", "emps"])
def test_algorithm1_invariants(name):
    pb = {"smo": experiments.smo_marginal, "toy": experiments.toy_marginal, "vehicle": experiments.vehicle_marginal,
          "emps": experiments.emps_marginal}[name](T=12)
    N = 80
    alg = marginal_oracle(pb, N)
    st, iv, sst, w, anc, ss, obs, ll = alg(CanonRand(SEED, N))
    T = pb.T
    assert st.shape == (T, N, pb.init_state_mean.size) and iv[0].shape == (T, N, 1) and w.shape == (T, N) and anc.shape == (T - 1, N)
    assert np.allclose(w.sum(axis=1), 1.0) and np.all(np.diff(anc, axis=1) >= 0)
    lam = pb.forgetting_factor
    # T3 is a deterministic count: sum_k lam^k (src/Algorithm1.py:317-320, BI:59)
    assert np.allclose(ss[0][3], sum(lam ** k for k in range(T)))
    assert np.allclose(sst[0][3], [sum(lam ** k for k in range(t + 1)) for t in range(T)])
    # per-particle statistics are those of the particle's own ancestral line
    b = N // 2
    line = [b]
    for t in range(T - 2, -1, -1):
        line.append(anc[t, line[-1]])
    line = line[::-1]
    T0 = np.zeros_like(ss[0][0][0]); T1 = np.zeros_like(ss[0][1][0])
    for t in range(T):
        phi = alg.basis_fcn[0](st[t, line[t]][None], alg.inputs[t])[0]
        T0 = lam * T0 * (t > 0) + np.outer(phi, iv[0][t, line[t]]) if t else np.outer(phi, iv[0][t, line[t]])
        T1 = lam * T1 + np.outer(phi, phi) if t else np.outer(phi, phi)
    assert np.allclose(ss[0][0][b], T0, rtol=1e-10, atol=1e-12) and np.allclose(ss[0][1][b], T1, rtol=1e-10, atol=1e-12)
    assert np.isfinite(ll).all() and obs.shape[:2] == (T, N)


def test_algorithm3_keeps_the_reference_and_consumes_its_statistics():
    pb = experiments.smo_marginal(T=10)
    N = 60
    alg = marginal_oracle(pb, N, "Algorithm3")
    ref_x, ref_iv = pb.X_true, [pb.int_var_true[0]]
    ref_stats = mo.trajectory_stats(alg, ref_x, ref_iv)
    rand = CanonRand(SEED, N)
    traj, ivt, tr = alg(rand, ref_x, ref_iv, ref_stats)
    assert np.allclose(tr["state_trace"][:, -1], ref_x)           # conditioned particle (src/Algorithm3.py:134, :221)
    assert traj.shape == (pb.T, 2) and ivt[0].shape == (pb.T,)
    # one step by hand: the remaining reference statistics shrink by exactly the consumed sample (:165-176)
    st, ivtr, _, lw, anc, ss = alg._init_algorithm(rand)
    rs = [tuple(np.asarray(r) for r in ref_stats[0])]
    out = alg.step(rand, 1, lw[0], st[0], [ivtr[0][0]], ss, ref_x[1], [ref_iv[0][1]], rs)
    phi = alg.basis_fcn[0](ref_x[1][None], alg.inputs[1])[0]
    assert np.allclose(out[5][0][1], rs[0][1] - np.outer(phi, phi)) and np.isclose(out[5][0][3], rs[0][3] - 1)
    assert np.array_equal(out[1][-1], ref_x[1]) and out[2][0][-1, 0] == ref_iv[0][1]


@pytest.mark.parametrize("name", ["smo", "toy", "vehicle"])
def test_restatement_matches_committed_vectors(name):
    """tests/golden/marginal_runs.json (tools/make_golden.py): regression vectors of the restatement on the canonical Philox streams."""
    import json
    import os

    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "marginal_runs.json")))[name]
    pb = {"smo": experiments.smo_marginal, "toy": experiments.toy_marginal, "vehicle": experiments.vehicle_marginal}[name](T=g["T"])
    N = g["N"]
    o1 = marginal_oracle(pb, N)(CanonRand(g["seed"], N))
    assert np.array_equal(o1[4], np.array(g["alg1_ancestors"]))
    assert np.allclose(o1[0][-1].reshape(-1), g["alg1_state_last"], rtol=1e-10, atol=1e-13)
    assert np.allclose(o1[3][-1], g["alg1_weights_last"], rtol=1e-8, atol=1e-14)
    for i in range(len(pb.basis)):
        assert np.allclose(o1[1][i][-1].reshape(-1), g["alg1_int_var_last"][i], rtol=1e-9, atol=1e-12)
        assert np.allclose(np.diag(o1[2][i][1][-1]), g["alg1_T1_trace_diag_last"][i], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("name", ["smo", "vehicle", "emps", "toy"])
def test_trajectory_basis_equals_per_step_calls(name):
    """Algorithm2's trajectory statistics evaluate the basis for all T rows in one call (row t with inputs[t]); that call must equal
    the reference's per-step calls (src/Algorithm2.py:81-93)."""
    mk = {"smo": lambda: experiments.smo_marginal(T=30), "vehicle": lambda: experiments.vehicle_marginal(T=30),
          "emps": lambda: experiments.emps_marginal(T=30), "toy": lambda: experiments.toy_marginal(T=20)}[name]
    pb = mk()
    X = np.asarray(pb.X_true, dtype=np.float64).reshape(pb.T, -1)
    U = np.asarray(pb.inputs, dtype=np.float64).reshape(pb.T, -1)
    for bf in pb.basis_fcn():
        assert hasattr(bf, "trajectory")
        whole = bf.trajectory(X, U)
        steps = np.vstack([bf(X[t:t + 1], U[t]) for t in range(pb.T)])
        assert whole.shape == steps.shape and np.array_equal(whole, steps)


def test_chi2_sampler_moments():
    """canon.chi2 (the draw PGAS.sample_params' Bartlett diagonal consumes on the device): mean nu, variance 2 nu."""
    for nu in (0.7, 3.0, 2001.0):
        c = canon.chi2(SEED, 18, 0, 0, np.full(40000, nu))
        assert abs(c.mean() - nu) < 5 * np.sqrt(2 * nu / 40000) and abs(c.var() - 2 * nu) < 0.1 * 2 * nu
        assert (c > 0).all()
