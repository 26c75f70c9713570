"""GPU tests of the MNIW sufficient statistics (fp64 MFMA SYRK) and of the PGAS Gibbs loop."""
import numpy as np
import pytest
import torch

from common import experiments, host_param_draws, numpy_csmc, pgas_amd, pgas_numpy

pytestmark = pytest.mark.gpu
# A T-term fp64 dot product carries a rounding error of at most ~T eps times the sum of the absolute products (eps = 1.1e-16), whatever
# the summation order (NumPy's blocked sums on one side, the split-K MFMA accumulation on the other): 1e-15 per row and unit of scale,
# i.e. 2e-12 relative at T = 2000 -- the bound DESIGN.md quotes for the statistics.
RTOL = 1e-15


def _abs_scales(pb, Phi):
    """max of |Phi|^T |X+|, |Phi|^T |Phi|, |X+|^T |X+|: what the rounding-error bound of each statistic is relative to."""
    Pa, Xp = np.abs(Phi), np.abs(np.asarray(pb.X_true, dtype=np.float64).reshape(pb.T, -1)[1:])
    return {"T0": (Pa.T @ Xp).max(), "T1": (Pa.T @ Pa).max(), "T2": (Xp.T @ Xp).max()}


def _phi_numpy(pb):
    nm = numpy_csmc(pb, 4)
    T = pb.T
    u = np.asarray(pb.inputs, dtype=np.float64).reshape(T, -1)
    return np.vstack([nm.basis(pb.X_true[t : t + 1], u[t]) for t in range(T - 1)])  # traj[:-1] with inputs[:-1] (Q3)


@pytest.mark.parametrize("maker", [lambda: experiments.smo_pgas(T=300), lambda: experiments.toy(T=40), lambda: experiments.emps_pgas(T=50),
                                   lambda: experiments.vehicle_pgas(T=60, M=27)])
def test_suffstats_match_numpy(maker):
    pb = maker()
    pg = pgas_amd.PGAS(256, 2, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
    T0, T1, T2, T3 = pg.cSMC.engine.suffstats(pb.X_true)
    Phi = _phi_numpy(pb)
    r0, r1, r2, r3 = pgas_numpy.suff_stats(pb.X_true, Phi)
    sc = _abs_scales(pb, Phi)
    for g, r, nm in ((T0, r0, "T0"), (T1, r1, "T1"), (T2, r2, "T2")):
        g = g.cpu().numpy()
        scale = sc[nm]
        assert np.abs(g - r).max() <= RTOL * scale * pb.T, f"{nm}: max |d| = {np.abs(g - r).max():.3e} (scale {scale:.3e})"
    assert T3 == r3
    assert np.allclose(T1.cpu().numpy(), T1.cpu().numpy().T, rtol=0, atol=1e-13 * np.abs(r1).max())


@pytest.mark.parametrize("maker,splits", [(lambda: experiments.emps_pgas(T=2000), 0), (lambda: experiments.emps_pgas(T=2000), 3),
                                          (lambda: experiments.smo_pgas(T=2000), 0), (lambda: experiments.smo_pgas(T=2000), 1)])
def test_suffstats_full_size(maker, splits):
    """T = 2000 rows, M = 729 / 41 columns: the LDS-staged split SYRK against numpy's Phi^T Phi, for the automatic and a forced row split."""
    pb = maker()
    pg = pgas_amd.PGAS(256, 2, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
    pg.cSMC.engine.set_option(10, splits)
    T0, T1, T2, T3 = pg.cSMC.engine.suffstats(pb.X_true)
    Phi = _phi_numpy(pb)
    r0, r1, r2, r3 = pgas_numpy.suff_stats(pb.X_true, Phi)
    sc = _abs_scales(pb, Phi)
    for g, r, nm in ((T0, r0, "T0"), (T1, r1, "T1"), (T2, r2, "T2")):
        g = g.cpu().numpy()
        assert g.shape == r.shape
        assert np.abs(g - r).max() <= RTOL * sc[nm] * pb.T, f"{nm}: max |d| = {np.abs(g - r).max():.3e}"
    assert torch.equal(T1, T1.T) and T3 == r3


@pytest.mark.parametrize("M,T", [(62, 37), (63, 37), (64, 37), (65, 21), (127, 18), (128, 18), (64, 2), (27, 3)])
def test_suffstats_block_boundaries(M, T):
    """[Phi | X+] column counts around the 64-wide SYRK block (X+ inside the last block, straddling two blocks, alone in a new block)
    and row counts below one 16-row panel."""
    pb = experiments.emps_pgas(T=T, M=M)
    pg = pgas_amd.PGAS(256, 2, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
    T0, T1, T2, T3 = pg.cSMC.engine.suffstats(pb.X_true)
    Phi = _phi_numpy(pb)
    r0, r1, r2, r3 = pgas_numpy.suff_stats(pb.X_true, Phi)
    sc = _abs_scales(pb, Phi)
    for g, r, nm in ((T0, r0, "T0"), (T1, r1, "T1"), (T2, r2, "T2")):
        g = g.cpu().numpy()
        assert g.shape == r.shape and np.abs(g - r).max() <= RTOL * max(sc[nm], 1e-300) * pb.T, f"{nm} (M={M}, T={T}): max |d| = {np.abs(g - r).max():.3e}"
    assert torch.equal(T1, T1.T) and torch.equal(T2, T2.T) and T3 == r3


def test_sample_params_matches_numpy_on_same_draws():
    pb = experiments.smo_pgas(T=200)
    pg = pgas_amd.PGAS(256, 2, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
    key = pgas_amd.random.key(99)
    draws = pg.param_draws(key)
    A, S = pg.sample_params(key, pb.X_true, draws=draws)
    draws = {k: v.cpu().numpy() for k, v in draws.items()}
    host = host_param_draws(key, pb.nx, pg.cSMC.engine.M, float(pb.GP_prior[3]) + (pb.T - 1))
    assert all(np.array_equal(draws[k], host[k]) for k in host)   # device Philox draws == canonical C oracle, bit for bit
    r0, r1, r2, r3 = pgas_numpy.suff_stats(pb.X_true, _phi_numpy(pb))
    prior = tuple(np.asarray(p) for p in pb.GP_prior)
    Ao, So, _ = pgas_numpy.sample_params(prior, r0, r1, r2, r3, draws["chi2"], draws["normals_T"], draws["normals_A"])
    assert np.allclose(S.cpu().numpy(), So, rtol=1e-9, atol=0)
    assert np.allclose(A.cpu().numpy(), Ao, rtol=1e-7, atol=1e-9 * np.abs(Ao).max())


def test_pgas_gibbs_loop_runs_and_conditions():
    pb = experiments.toy(T=40)
    K = 6
    pg = pgas_amd.PGAS(512, K, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
    trace, ll = pg(pgas_amd.random.key(12345678), pb.X_true)
    assert trace.shape == (pb.T, K, 1) and ll.shape == (pb.T, K)
    assert torch.equal(trace[:, 0, 0].cpu(), torch.as_tensor(pb.X_true[:, 0]))
    assert bool(torch.isfinite(trace).all()) and bool(torch.isfinite(ll).all())
    # log-likelihood definition (src/PGAS.py:383-392)
    y = np.asarray(pb.observations).reshape(pb.T)
    expect = -0.5 * np.log(2 * np.pi * 4.0) - 0.5 * (y[:, None] - trace[:, :, 0].cpu().numpy()) ** 2 / 4.0
    assert np.allclose(ll.cpu().numpy(), expect, rtol=1e-12, atol=1e-12)
