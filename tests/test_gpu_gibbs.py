"""Whole Gibbs sampler on the device (PGAS.__call__, src/PGAS.py:345-397) on the reference's Toy problem."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _driver(name="Toy_Example_Simulation"):
    import sys

    ex = os.path.join(ROOT, "examples")
    if ex not in sys.path:
        sys.path.insert(0, ex)   # the drivers share examples/_marginal_driver.py
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("corrected", [False, True])
def test_toy_posterior_mean_approaches_true_function(corrected):
    """SURVEY 8c KAT 7: the Toy PGAS posterior mean of A phi(x) moves from the prior (zero) towards 10 sinc(x/7) on the data
    range.  Statistical test, and a weak one by nature: 39 transitions, process and measurement noise variance 4 each.
    Calibration (tools/toy_probe.py and the literal NumPy restatement run through the same Gibbs loop with NumPy's RNG):
    RMSE on the central data range 3.67 (this engine), 3.68 (NumPy restatement), 1.10 given the TRUE states, 6.3 for the
    prior mean -- the engine reproduces the reference algorithm's statistics, which is what is asserted."""
    res = _driver().run(iterations=300, particles=200, resample_before_propagate=corrected)
    assert res["pgas_Sigma_X"].shape == (40, 300, 1) and res["pgas_log_likelihood"].shape == (40, 300)
    assert np.isfinite(res["pgas_Sigma_X"]).all() and np.isfinite(res["pgas_log_likelihood"]).all()
    X = res["X"][:, 0]
    lo, hi = np.percentile(X, 10), np.percentile(X, 90)
    near = (res["x_plot"] > lo) & (res["x_plot"] < hi)
    err = res["pgas_fcn_mean"][near] - res["fx_true_plot"][near]
    rmse = float(np.sqrt(np.mean(err ** 2)))
    print("toy posterior RMSE on the central data range:", rmse, "corrected" if corrected else "reference mode")
    rmse_prior = float(np.sqrt(np.mean(res["fx_true_plot"][near] ** 2)))
    assert rmse < 4.5 and rmse < 0.75 * rmse_prior
    # the sampled trajectories follow the data: the posterior-mean trajectory correlates with the simulated truth
    xm = res["pgas_Sigma_X"][1:, 100:, 0].mean(axis=1)
    corr = float(np.corrcoef(xm, X[1:])[0, 1])
    print("correlation of the posterior-mean trajectory with the truth:", corr)
    assert corr > 0.5


def test_emps_driver_fields_and_tracking():
    """examples/EMPS_Simulation.py (PGAS part of the reference driver, synthetic data): field names and shapes of the .mat
    dictionary (EMPS_Simulation.py:128-160) and a sanity bound on the sampled positions (measurement noise std 1e-2)."""
    drv = _driver("EMPS_Simulation")
    res = drv.run(iterations=4, particles=256, steps=300)
    ppb = res.pop("_pb")
    assert res["offline_Sigma_X_PGAS"].shape == (300, 4, 2) and res["offline_log_likelihood_PGAS"].shape == (300, 4)
    assert res["PGAS_mean"].shape == (2, 729) and res["PGAS_T1"].shape == (729, 729) and res["PGAS_T3"] == res["prior_T3_PGAS"] + 299
    marg, mpb = drv.run_marginal(iterations=3, particles=200, steps=300, log=lambda *_: None)
    assert marg["online_Sigma_F"].shape == (300, 200, 1) and marg["offline_T1"].shape == (3, 9, 9) and marg["offline_mean"].shape == (1, 9)
    assert np.isfinite(marg["online_Sigma_X"]).all() and np.isfinite(marg["offline_Sigma_X"]).all()
    r_alg2, r_pgas = drv.validation_rmse(marg["offline_mean"], res["PGAS_mean"], mpb, ppb, steps=120)
    assert np.isfinite(r_alg2) and np.isfinite(r_pgas)
    assert np.isfinite(res["offline_Sigma_X_PGAS"]).all()
    err = res["offline_Sigma_X_PGAS"][:, -1, 0] - res["X"][:, 0]
    assert np.sqrt(np.mean(err ** 2)) < 0.05


def test_smo_driver_learns_the_spring_damper_force():
    """examples/SingleMassOscillator_Simulation.py (BASELINE configs[0], shortened): the reference's .mat fields, and the online
    estimate of the latent force F_sd = F_spring + F_damper is closer to the truth than the zero prior mean."""
    drv = _driver("SingleMassOscillator_Simulation")
    res = drv.run(particles=200, iterations=3, steps=400, log=lambda *_: None)
    T, N, K = 400, 200, 3
    assert res["online_Sigma_X"].shape == (T, N, 2) and res["online_Sigma_F"].shape == (T, N, 1) and res["online_weights"].shape == (T, N)
    assert res["online_T0"].shape == (T, 41, 1) and res["online_T1"].shape == (T, 41, 41) and res["online_T3"].shape == (T,)
    assert res["offline_Sigma_X"].shape == (T, K, 2) and res["offline_Sigma_F"].shape == (T, K, 1) and res["offline_T1"].shape == (K, 41, 41)
    for k in ("online_Sigma_X", "online_Sigma_F", "offline_Sigma_X", "offline_log_likelihood"):
        assert np.isfinite(res[k]).all(), k
    # filtered position follows the simulated truth (measurement noise std 0.03)
    xm = (res["online_Sigma_X"][:, :, 0] * res["online_weights"]).sum(axis=1)
    assert np.sqrt(np.mean((xm[50:] - res["X"][50:, 0]) ** 2)) < 0.05
    rmse, rms_true = drv.posterior_force_rmse(res, "online")
    print("online F_sd RMSE", rmse, "RMS of the true force", rms_true)
    assert rmse < 0.6 * rms_true


def test_vehicle_driver_two_latent_functions():
    """examples/VehicleSimulation_Simulation.py (shortened): the reference's .mat fields for both tyres, finite outputs, and the
    filtered yaw rate follows the simulated truth."""
    drv = _driver("VehicleSimulation_Simulation")
    res = drv.run(particles=200, iterations=3, steps=300, log=lambda *_: None)
    T, N, K = 300, 200, 3
    for s in "fr":
        assert res[f"online_Sigma_mu_{s}"].shape == (T, N, 1) and res[f"online_T1_{s}"].shape == (T, 20, 20)
        assert res[f"offline_Sigma_mu_{s}"].shape == (T, K, 1) and res[f"offline_T0_{s}"].shape == (K, 20, 1)
        assert res[f"online_Sigma_alpha_{s}"].shape == (T, N) and res[f"offline_Sigma_alpha_{s}"].shape == (T, K)
    assert res["online_Sigma_Y"].shape == (T, N, 2) and res["offline_Sigma_X"].shape == (T, K, 2)
    for k in ("online_Sigma_X", "online_Sigma_mu_f", "online_Sigma_mu_r", "offline_Sigma_X", "offline_log_likelihood", "online_T1_r"):
        assert np.isfinite(res[k]).all(), k
    xm = (res["online_Sigma_X"][:, :, 0] * res["online_weights"]).sum(axis=1)
    assert np.sqrt(np.mean((xm[20:] - res["X"][20:, 0]) ** 2)) < 0.05


@pytest.mark.parametrize("name,N,K", [("toy", 700, 4), ("smo", 1500, 4)])
def test_pgas_chain_against_restated_chain(name, N, K):
    """PGAS.__call__ (src/PGAS.py:345-397) against oracle/pgas_numpy.pgas_chain on identical randomness: the key splits are
    recomputed here from the root key, the parameter draws come from the same Philox streams, and every sweep of the restated
    chain is the canonical C oracle run with the DEVICE's (A_k, S_k) (a sweep is only reproducible bit for bit from identical
    parameters; the restatement's own draws are compared with the device's at 1e-9).  state_trace must be equal bit for bit."""
    import torch

    from common import canon_model, experiments, host_param_draws, numpy_csmc, pgas_amd
    from oracle import pgas_numpy as o
    from pgas_amd import random as prng

    pb = experiments.toy(T=30) if name == "toy" else experiments.smo_pgas(T=25)
    root = 20241004
    pg = pgas_amd.PGAS(N, K, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
    trace, ll = pg(root, pb.X_true)
    assert tuple(trace.shape) == (pb.T, K, pb.nx) and tuple(ll.shape) == (pb.T, K)
    # the reference's key handling (:356, :365, :377), recomputed independently of PGAS.__call__'s bookkeeping
    key, key_para = prng.split(root, 2)
    para_keys, step_keys = [key_para], [None]
    for k in range(1, K):
        key, ks = prng.split(key, 2)
        key, kp = prng.split(key, 2)
        step_keys.append(ks)
        para_keys.append(kp)
    assert para_keys == pg.chain_log["para_keys"] and step_keys == pg.chain_log["step_keys"]
    # the parameter draws are generated on the device; the restatement gets them recomputed on the host from the same keys
    df = float(pb.GP_prior[3]) + (pb.T - 1)
    draws = [host_param_draws(kp, pb.nx, pg.cSMC.engine.M, df) for kp in para_keys]
    for kp, d in zip(para_keys, draws):
        dd = pg.param_draws(kp)
        assert all(np.array_equal(dd[n].cpu().numpy(), d[n]) for n in d), "device parameter draws differ from the canonical streams"
    dev_params = [(A.cpu().numpy(), S.cpu().numpy()) for A, S in pg.chain_log["params"]]

    cm = canon_model(pb, N)
    L0 = np.linalg.cholesky(pb.init_state_cov)

    def sweep(seed, ref, A, S):
        # the chain keeps error_cov on the device and factors it there (pgas_set_params_dev); chol_parts_dev is that factorisation
        LS, LSinv, cS = cm.chol_parts_dev(S)
        return cm.sweep(seed, ref, A, LS, LSinv, cS, pb.init_state_mean, L0)[0]

    nc = numpy_csmc(pb, N)   # literal NumPy callables (basis, likelihood) built from the reference formulas
    prior = tuple(np.asarray(g, dtype=np.float64) if np.ndim(g) else float(g) for g in pb.GP_prior)
    st, llo, own = o.pgas_chain(sweep, nc.basis, nc.lik, prior, pb.observations, pb.inputs, pb.X_true, K, step_keys, draws, params=dev_params)
    assert np.array_equal(trace.cpu().numpy(), st), "state_trace differs from the restated chain"
    np.testing.assert_allclose(ll.cpu().numpy(), llo, rtol=1e-12, atol=1e-12)
    for k, ((A, S), (Ao, So)) in enumerate(zip(dev_params, own)):
        np.testing.assert_allclose(A, Ao, rtol=1e-9, atol=1e-9 * np.abs(Ao).max(), err_msg=f"coeff_mat of iteration {k}")
        np.testing.assert_allclose(S, So, rtol=1e-9, atol=1e-12, err_msg=f"error_cov of iteration {k}")
    assert not np.array_equal(st[:, 0], st[:, K - 1]), "the chain must move"
    # what the kernels ran the last sweep with is exactly that factorisation (and NumPy's to rounding)
    LSd, LSinvd, cSd = pg.cSMC.engine.get_params()
    A_last, S_last = pg.chain_log["params"][K - 1]
    pg.cSMC.engine.set_params(A_last, S_last)
    LSd, LSinvd, cSd = pg.cSMC.engine.get_params()
    LSo, LSinvo, cSo = cm.chol_parts_dev(S_last.cpu().numpy())
    assert np.array_equal(LSd, LSo) and np.array_equal(LSinvd, LSinvo) and cSd == cSo
    LSn, LSinvn, cSn = cm.chol_parts(S_last.cpu().numpy())
    np.testing.assert_allclose(LSd, LSn, rtol=1e-14, atol=0)
    np.testing.assert_allclose(LSinvd, LSinvn, rtol=1e-13, atol=0)
    assert abs(cSd - cSn) <= 1e-14 * abs(cSn)


def test_gibbs_iterations_make_no_host_round_trip():
    """A Gibbs iteration (sample_params -> set_params -> sweep, src/PGAS.py:360-378) enqueues work only: with device synchronisation made an
    error (torch.cuda.set_sync_debug_mode) the chain still runs, i.e. no .cpu() / .item() / blocking copy hides in the loop."""
    import torch

    from common import experiments, pgas_amd

    pb = experiments.toy(T=30)
    pg = pgas_amd.PGAS(500, 3, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
    pg(7, pb.X_true)   # first call: allocations (those may synchronise)
    torch.cuda.synchronize()
    dev = pg.cSMC.device
    ref = torch.as_tensor(pb.X_true, device=dev).reshape(pb.T, -1)
    A, S = pg.sample_params(11, ref)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for k in range(3):
            traj = pg.cSMC(100 + k, ref, A, S)
            A, S = pg.sample_params(200 + k, traj.reshape(pb.T, -1))
            ref = traj.reshape(pb.T, -1)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(A).all()) and bool(torch.isfinite(S).all())

