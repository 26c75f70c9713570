"""Known-answer tests pinning oracle/pgas_numpy.py (SURVEY.md 8c list): everything that can be
derived without running the (un-importable) JAX reference, with SciPy as an independent check."""
import json
import os

import numpy as np
import scipy.stats as sst

from oracle import pgas_numpy as o

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_systematic_resampling_kat():
    assert o.systematic_SISR(0.5, [0.1, 0.2, 0.3, 0.4]).tolist() == [1, 2, 3, 3]
    assert o.systematic_SISR(0.3, np.zeros(7)).tolist() == list(range(7))            # Filtering.py:25 fallback
    assert o.systematic_SISR(0.5, [-1.0, 0.5, 0.5]).tolist() == o.systematic_SISR(0.5, [0.0, 0.5, 0.5]).tolist()
    rng = np.random.default_rng(3)
    w = rng.random(1000)
    w /= w.sum()
    idx = o.systematic_SISR(rng.random(), w)
    assert np.all(np.diff(idx) >= 0)
    assert np.all(np.abs(np.bincount(idx, minlength=1000) - 1000 * w) <= 1 + 1e-9)


def test_reconstruct_trajectory_hand_built():
    P = np.arange(12, dtype=float).reshape(4, 3)           # P[t, n] = 3 t + n
    anc = np.array([[2, 0, 1], [1, 1, 0], [0, 2, 2]])
    # idx 1 at t=3 -> anc[2,1]=2 at t=2 -> anc[1,2]=0 at t=1 -> anc[0,0]=2 at t=0
    assert o.reconstruct_trajectory(P[:, :, None], anc, 1).tolist() == [2.0, 3.0, 8.0, 10.0]


def test_basis_index_tables_golden():
    g = json.load(open(os.path.join(GOLD, "basis_index_tables.json")))
    for name, spec in g.items():
        S, _, _ = o.hilbert_index_table(spec["num_fcn"], spec["domain"], spec.get("idx_start", 1), spec.get("idx_step", 1))
        assert S.astype(int).tolist()[: len(spec["first"])] == spec["first"], name
        assert S.max(axis=0).astype(int).tolist() == spec["max"], name
        assert len(S) == spec["num_fcn"]
        assert len({tuple(r) for r in S.astype(int).tolist()}) == spec["num_fcn"], "indices must be distinct"


def test_basis_orthonormal_and_dirichlet():
    phi, sd, S = o.generate_Hilbert_BasisFunction(6, np.array([[-1.0, 2.0], [0.0, 4.0]]), 0.7, 3.0)
    # Gauss-Legendre quadrature of phi_i phi_j over the box
    n = 60
    gx, wx = np.polynomial.legendre.leggauss(n)
    x0 = 0.5 * 3.0 * gx + 0.5
    x1 = 0.5 * 4.0 * gx + 2.0
    Gm = np.zeros((6, 6))
    for a, wa in zip(x0, wx):
        for b, wb in zip(x1, wx):
            f = phi(np.array([a, b]))
            Gm += wa * wb * 1.5 * 2.0 * np.outer(f, f)
    assert np.allclose(Gm, np.eye(6), atol=1e-10)
    for pt in ([-1.0, 1.0], [2.0, 3.0], [0.3, 0.0], [0.3, 4.0]):
        assert np.allclose(phi(np.array(pt)), 0.0, atol=1e-14)


def test_spectral_density_closed_form_and_ranges():
    assert np.isclose(o.spectral_density_Gaussian(np.array([0.0, 0.0]), 2.0, 3.0), 2.0 * (2 * np.pi) * 9.0)
    g = json.load(open(os.path.join(GOLD, "basis_index_tables.json")))
    for name, spec in g.items():
        if "sd_range" not in spec:
            continue
        _, sd, _ = o.generate_Hilbert_BasisFunction(spec["num_fcn"], np.array(spec["domain"]), spec["lengthscale"], spec["scale"],
                                                    spec.get("idx_start", 1), spec.get("idx_step", 1))
        assert np.isclose(sd.min(), spec["sd_range"][0], rtol=2e-3) and np.isclose(sd.max(), spec["sd_range"][1], rtol=2e-3), name


def test_mniw_round_trip_and_statistics():
    rng = np.random.default_rng(5)
    M, n = 7, 2
    mean = rng.standard_normal((n, M))
    V = rng.standard_normal((M, M)); V = V @ V.T + M * np.eye(M)
    Psi = rng.standard_normal((n, n)); Psi = Psi @ Psi.T + n * np.eye(n)
    back = o.prior_mniw_2naturalPara_inv(*o.prior_mniw_2naturalPara(mean, V, Psi, 5))
    for a, b in zip(back[:3], (mean, V, Psi)):
        assert np.allclose(a, b, rtol=1e-10, atol=1e-10)
    assert back[3] == 5
    Phi, X = rng.standard_normal((20, M)), rng.standard_normal((21, n))
    acc = [0, 0, 0, 0]
    for t in range(20):
        st = o.prior_mniw_calcStatistics(X[t + 1], Phi[t])
        acc = [a + s for a, s in zip(acc, st)]
    T0, T1, T2, T3 = o.suff_stats(X, Phi)
    assert np.allclose(acc[0], T0) and np.allclose(acc[1], T1) and np.allclose(acc[2], T2) and acc[3] == T3 == 20
    assert np.allclose(o.prior_mniw_mean(*o.prior_mniw_2naturalPara(mean, V, Psi, 5)[:2]), mean)


def test_mvn_logpdf_and_log_base_measure_against_scipy():
    rng = np.random.default_rng(6)
    C = rng.standard_normal((3, 3)); C = C @ C.T + np.eye(3)
    mu, x = rng.standard_normal((5, 3)), rng.standard_normal(3)
    ref = np.array([sst.multivariate_normal.logpdf(x, m, C) for m in mu])
    assert np.allclose(o.mvn_logpdf(x, mu, C), ref, rtol=1e-12)
    T1 = C
    T0 = rng.standard_normal((3, 2))
    T2 = T0.T @ np.linalg.solve(T1, T0) + np.eye(2)
    v = o.prior_mniw_log_base_measure(T0, T1, T2, 7.0)
    # det(Psi) = 1 by construction
    from scipy.special import multigammaln
    expect = -0.5 * 2 * 3 * np.log(2 * np.pi) + 0.5 * 2 * np.log(np.linalg.det(T1)) - 0.5 * 7 * 2 * np.log(2) - multigammaln(3.5, 2)
    assert np.isclose(v, expect, rtol=1e-10)


def test_sample_params_inverse_wishart_mean():
    """Moment check of the Bartlett construction.

    Quirk Q14 (found by this test): the reference forms L = chol(Psi)^-1 and C = L T (src/PGAS.py:317-332), so
    W = C C^T ~ Wishart(df, L L^T) with L L^T = (Lc^T Lc)^-1, not Psi^-1 = (Lc Lc^T)^-1.  Hence
    S ~ IW(df, Lc^T Lc) and E[S] = Lc^T Lc / (df - p - 1); this equals Psi/(df - p - 1) only for diagonal Psi.
    Reproduced as is (the product's sample_params uses the same algebra)."""
    rng = np.random.default_rng(7)
    M, n = 3, 2
    prior = o.prior_mniw_2naturalPara(np.zeros((n, M)), np.eye(M), np.array([[2.0, 0.3], [0.3, 1.0]]), 12.0)
    acc = np.zeros((n, n))
    K = 4000
    for _ in range(K):
        chi2 = rng.chisquare(12.0 - np.arange(n))
        _, S, (mean, col, Psi, df) = o.sample_params(prior, 0, 0, 0, 0, chi2, rng.standard_normal((n, n)), rng.standard_normal((n, M)))
        acc += S
    Lc = np.linalg.cholesky(Psi)
    assert np.allclose(acc / K, Lc.T @ Lc / (df - n - 1), rtol=0.08)
    assert not np.allclose(acc / K, Psi / (df - n - 1), rtol=0.08)  # the textbook mean is NOT what the reference samples


def test_conditional_smc_invariants():
    """x_new[N-1] == ref_t, a[N-1] == ref_idx, logw_new = l(x_new) - l_aux[a], Q1 (no state gather)."""
    from common import canon_rand, experiments, numpy_csmc

    pb = experiments.smo_pgas(T=6)
    N = 64
    nm = numpy_csmc(pb, N)
    A, S = experiments.initial_params(pb)
    rand = canon_rand(11, N, pb.T, 2)
    x0 = nm.init_state(rand["z0"], pb.X_true[0])
    lw, xn, a = nm.step(rand["u_resample"][1], rand["u_ancestor"][1], rand["z"][1], 1, np.zeros(N), x0, A, S, pb.X_true[1])
    assert np.array_equal(xn[-1], pb.X_true[1])
    aux = nm.basis(x0, nm.u[1]) @ A.T
    assert np.allclose(xn[:-1], aux[:-1] + rand["z"][1][:-1] @ np.linalg.cholesky(S).T)        # from x0[i], not x0[a[i]]
    ll_aux = nm.lik(nm.y[1], aux, nm.u[1])
    assert np.allclose(lw, nm.lik(nm.y[1], xn, nm.u[1]) - ll_aux[a])
    w_anc = o.softmax(ll_aux + o.mvn_logpdf(pb.X_true[1], aux, S))
    assert a[-1] == min(np.searchsorted(np.cumsum(w_anc), rand["u_ancestor"][1]), N - 1)


def test_pgas_chain_restatement_structure():
    """oracle/pgas_numpy.pgas_chain (src/PGAS.py:345-397): with a sweep that returns a known function of its inputs, the
    restated chain must thread the trajectories, the parameters and the draws exactly as the reference loop does:
    trace[0] = init_ref, sweep k sees trace[k-1] and the parameters drawn from trace[k-1], the likelihood block is evaluated on the
    swapped (T,K,nx) trace, and `params` teacher-forces the sweeps without changing the restatement's own draws."""
    from oracle import pgas_numpy as o

    rng = np.random.default_rng(0)
    T, nx, M, K = 9, 1, 4, 4
    y = rng.standard_normal(T)
    u = np.zeros(T)
    ref = rng.standard_normal((T, nx))
    basis = lambda x, ut: np.cos(np.arange(1, M + 1)[None] * np.atleast_2d(x)[:, :1])   # noqa: E731
    lik = lambda obs, x, ut: -0.5 * (obs[0] - np.atleast_2d(x)[:, 0]) ** 2               # noqa: E731
    prior = o.prior_mniw_2naturalPara(np.zeros((nx, M)), np.eye(M), np.eye(nx), 3.0)
    draws = [dict(chi2=rng.chisquare(5, nx), normals_T=rng.standard_normal((nx, nx)), normals_A=rng.standard_normal((nx, M))) for _ in range(K)]
    calls = []

    def sweep(seed, refk, A, S):
        calls.append((seed, refk.copy(), A.copy(), S.copy()))
        return refk + 0.1 * seed + A.sum()

    seeds = [None, 1, 2, 3]
    st, ll, own = o.pgas_chain(sweep, basis, lik, prior, y, u, ref, K, seeds, draws)
    assert st.shape == (T, K, nx) and ll.shape == (T, K) and len(own) == K and len(calls) == K - 1
    assert np.array_equal(st[:, 0], ref)
    for k in range(1, K):
        seed, refk, A, S = calls[k - 1]
        assert seed == seeds[k] and np.array_equal(refk, st[:, k - 1]) and np.array_equal(A, own[k - 1][0]) and np.array_equal(S, own[k - 1][1])
        assert np.array_equal(st[:, k], refk + 0.1 * seed + A.sum())
    # parameters of iteration k are the sample_params of trajectory k with draws[k]
    for k in range(K):
        Phi = np.vstack([basis(st[t:t + 1, k], u[t]) for t in range(T - 1)])
        A, S, _ = o.sample_params(prior, *o.suff_stats(st[:, k], Phi), draws[k]["chi2"], draws[k]["normals_T"], draws[k]["normals_A"])
        assert np.array_equal(A, own[k][0]) and np.array_equal(S, own[k][1])
    assert np.allclose(ll, -0.5 * (y[:, None] - st[:, :, 0]) ** 2)
    # teacher forcing: the sweeps see the forced parameters, the restatement's own draws still follow its trajectories
    forced = [(np.full((nx, M), 0.5), np.eye(nx)) for _ in range(K)]
    calls.clear()
    st2, _, own2 = o.pgas_chain(sweep, basis, lik, prior, y, u, ref, K, seeds, draws, params=forced)
    assert all(np.array_equal(c[2], forced[0][0]) for c in calls)
    assert np.array_equal(own2[0][0], own[0][0]) and not np.array_equal(st2[:, 1], st[:, 1])
