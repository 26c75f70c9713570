"""Host logic of the package (no GPU): basis setup, descriptors, MNIW algebra, keys."""
import numpy as np

import pgas_amd
from oracle import pgas_numpy as o
from pgas_amd import experiments
from pgas_amd import random as prng


def test_generate_hilbert_basis_matches_restatement():
    for args in ((41, np.array([[-7.5, 7.5], [-7.5, 7.5]]), 15 / 41, 100), (729, np.array([[-1, 1]] * 3), 0.5 / 729, 20),
                 (40, np.array([-30, 30]), 3, 50), (20, np.array([-0.5, 0.5]), 0.1, 2.0, 2, 2)):
        b, sd = pgas_amd.generate_Hilbert_BasisFunction(*args)
        phi, sd2, S = o.generate_Hilbert_BasisFunction(*args)
        assert np.array_equal(b.indices, S.astype(np.int32))
        assert np.allclose(sd, sd2, rtol=1e-14)
        x = np.full(b.D, 0.123)
        assert np.allclose(b(x), phi(x if b.D > 1 else 0.123), rtol=1e-12, atol=1e-15)


def test_basis_map_tables():
    b, _ = pgas_amd.generate_Hilbert_BasisFunction(27, np.array([[-1, 1]] * 3), 0.1, 1.0)
    bm = b.on([0, 1, 2], div=[0.4, 0.4, 160])
    state, inp = np.array([0.1, -0.2]), np.array([35.0])
    v = np.concatenate([state, inp])
    r = v[bm.sel] * bm.alpha + bm.beta
    direct = b.norm * np.prod(np.sin(np.pi * b.indices * r), axis=1)
    assert np.allclose(bm(state, inp), direct, rtol=1e-12)


def test_gaussian_likelihood_is_mvn_logpdf():
    import scipy.stats as sst

    lik = pgas_amd.GaussianLikelihood(np.array([[1.0, 0.5], [0.0, 2.0]]), np.array([[0.3, 0.1], [0.1, 0.2]]))
    x, y = np.array([0.2, -0.4]), np.array([0.1, 0.3])
    assert np.isclose(lik(y, x), sst.multivariate_normal.logpdf(y, lik.H @ x, lik.R))
    l1 = pgas_amd.GaussianLikelihood.of_component(0, 2, np.array([[1e-3]]))
    assert np.isclose(l1(0.05, x), sst.norm.logpdf(0.05, 0.2, np.sqrt(1e-3)))


def test_mniw_host_functions_match_restatement():
    rng = np.random.default_rng(1)
    M, n = 6, 2
    mean = rng.standard_normal((n, M))
    V = np.diag(rng.random(M) + 0.5)
    a = pgas_amd.prior_mniw_2naturalPara(mean, V, np.eye(n), 3)
    b = o.prior_mniw_2naturalPara(mean, V, np.eye(n), 3)
    for x, y in zip(a[:3], b[:3]):
        assert np.allclose(x, y, rtol=1e-13)
    ai = pgas_amd.prior_mniw_2naturalPara_inv(*a)
    assert np.allclose(ai[0], mean) and np.allclose(ai[1], V) and np.allclose(ai[2], np.eye(n))
    assert np.allclose(pgas_amd.prior_mniw_mean(a[0], a[1]), mean)
    st = pgas_amd.prior_mniw_calcStatistics(np.array([1.0, 2.0]), np.arange(3.0))
    assert st[0].shape == (3, 2) and st[1].shape == (3, 3) and st[2].shape == (2, 2) and st[3] == 1


def test_keys_and_host_student_t():
    """Key handling, and the one host-side sampler left in the package: Student-t variates computed by the library's own arithmetic on
    the CPU (pgas_m_rng_student_t_host) -- identical to the canonical C oracle's, which is what the device kernel reproduces
    (tests/test_gpu_marginal.py) -- with the right distribution."""
    from oracle import canon
    from pgas_amd._lib import student_t_host

    k = prng.key(12345678)
    a, b = prng.split(k, 2)
    assert a != b and prng.split(k, 2) == [a, b]
    nu = np.full(4000, 5.0)
    t = student_t_host(a, prng.STREAM_INTVAR, 3, nu)
    assert np.array_equal(t, canon.student_t(a, prng.STREAM_INTVAR, 3, 0, nu))
    assert abs(t.mean()) < 0.1 and abs(t.var() / (5.0 / 3.0) - 1) < 0.25   # Var t_5 = 5/3
    assert not hasattr(prng, "normal") and not hasattr(prng, "chisquare"), "normal / chi^2 variates come from the library, not from a second host sampler"


def test_experiment_definitions():
    pb = experiments.smo_pgas(T=60)
    assert pb.basis_fcn.basis.M == 41 and pb.GP_prior[0].shape == (41, 2) and pb.GP_prior[1].shape == (41, 41)
    assert pb.inputs[0] > 0 and pb.inputs[-1] < 0 and pb.inputs[30] == 0.0
    e = experiments.emps_pgas(T=200)
    assert e.basis_fcn.basis.M == 729 and np.abs(e.X_true[:, 0]).max() < 0.4 and np.abs(e.inputs).max() < 160
    t = experiments.toy()
    assert t.inputs.shape == (40, 0) and t.basis_fcn.basis.M == 40
    v = experiments.vehicle_pgas(T=300)
    assert v.basis_fcn.basis.M == 729 and v.observations.shape == (300, 2) and v.inputs.shape == (300, 2)
    assert np.all(v.inputs[:, 1] == 11.0) and np.abs(v.inputs[:, 0]).max() < 0.25


def test_predictive_and_log_base_measure_match_restatement():
    """BI:64-89 and :111-124 (only used by the drivers' post-processing and by Algorithm1/3)."""
    rng = np.random.default_rng(11)
    M, n = 9, 2
    mean = rng.standard_normal((n, M))
    V = rng.standard_normal((M, M)); V = V @ V.T + M * np.eye(M)
    Psi = np.array([[2.0, 0.3], [0.3, 1.0]])
    basis = rng.standard_normal((5, M))
    got = pgas_amd.prior_mniw_Predictive(mean, V, Psi, 7, basis)
    ref = o.prior_mniw_Predictive(mean, V, Psi, 7, basis)
    for g, r in zip(got, ref):
        assert np.allclose(g, r, rtol=1e-13)
    assert got[3] == 7 + 1 - n and np.allclose(got[1], basis @ V @ basis.T + np.eye(5))
    eta = pgas_amd.prior_mniw_2naturalPara(mean, V, Psi, 7)
    assert np.isclose(pgas_amd.prior_mniw_log_base_measure(*eta), o.prior_mniw_log_base_measure(*eta), rtol=1e-12)


def test_draw_pred_is_a_scaled_student_t():
    k = prng.key(99)
    row, col, df = np.array([[4.0]]), np.array([[2.25]]), 5
    d = np.array([pgas_amd.prior_mniw_drawPred(s, np.array([1.5]), col, row, df) for s in prng.split(k, 4000)]).reshape(-1)
    # mean 1.5, variance row * col * df / (df - 2)
    assert abs(d.mean() - 1.5) < 0.2
    assert abs(d.var() / (4.0 * 2.25 * df / (df - 2)) - 1) < 0.25
    assert np.array_equal(pgas_amd.prior_mniw_drawPred(k, np.array([1.5]), col, row, df), pgas_amd.prior_mniw_drawPred(k, np.array([1.5]), col, row, df))


