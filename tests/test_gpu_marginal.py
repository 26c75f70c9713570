"""GPU tests of the marginalised family (pgas_amd.Algorithm1 / Algorithm3 / Algorithm2, include/pgas_marginal.h).

Kernel numerics: each HIP kernel against a plain torch fp64 reference of the same operation (Cholesky-based quantities to
1e-10 relative, the statistics update exactly), the random-number kernels bit for bit against the canonical C oracle.
Algorithm parity: against the NumPy restatement of the reference (oracle/marginal_numpy.py) driven by the SAME Philox streams
(common.CanonRand), tolerance 1e-8 relative on continuous outputs, ancestor indices equal."""
import numpy as np
import pytest
import torch

from common import CanonRand, canon, experiments, marginal_oracle, pgas_amd
from oracle import marginal_numpy as mo

pytestmark = pytest.mark.gpu
SEED = 12345678


def _ops(N):
    from pgas_amd._lib import MarginalOps

    return MarginalOps(N)


def test_rng_kernels_bit_exact():
    N = 5000
    ops = _ops(N)
    for ncol in (1, 2, 3):
        z = ops.normal(SEED, 17, 9, ncol).cpu().numpy()
        assert np.array_equal(z, canon.normals(SEED, 17, 9, 0, N, ncol))
    nu = np.concatenate([np.full(1000, 1.0), np.full(1000, 3.0), np.linspace(2.0, 2000.0, 3000)])
    t = ops.student_t(SEED, 32, 4, torch.as_tensor(nu)).cpu().numpy()
    assert np.array_equal(t, canon.student_t(SEED, 32, 4, 0, nu))
    from pgas_amd._lib import student_t_host
    assert np.array_equal(t, student_t_host(SEED, 32, 4, nu)), "host-side helper draws (prior_mniw_drawPred) and device draws are one sampler"
    assert ops.uniform(SEED, 18, 7) == canon.uniform(SEED, 18, 7)


@pytest.mark.parametrize("N,M", [(130, 63), (70, 64), (90, 100), (257, 126), (64, 127)])
def test_mniw_solve_wide_bases(N, M):
    """63 <= M <= 126 (the reference's formulas, BI:48-50,64-124, are general in M): the two-rows-per-lane kernels; M = 127 is a clean error."""
    ops = _ops(N)
    if M > 126:
        z = torch.zeros(N, M, dtype=torch.float64, device=ops.device)
        from pgas_amd._lib import PgasError
        with pytest.raises(PgasError, match="outside"):
            ops.mniw_solve(z[0], torch.eye(M, dtype=torch.float64, device=ops.device), z, torch.zeros(N, M, M, dtype=torch.float64, device=ops.device), phi=z)
        return
    _check_mniw_solve(ops, N, M)


@pytest.mark.parametrize("N,M,nv", [(200, 41, 2), (130, 20, 3), (65, 100, 2), (64, 119, 8), (50, 7, 4)])
def test_mniw_kernels_with_several_components(N, M, nv):
    """Interface variables with nv > 1 components (eta0 (M, nv); BI:18-108 are general in n): solve, stored-factor solve, statistics
    update and weighted reduction against torch.linalg / einsum."""
    ops = _ops(N)
    g = torch.Generator(device="cpu").manual_seed(5)
    dev = ops.device
    rnd = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float64)   # noqa: E731
    B = rnd(N, M, 3)
    T1 = (B @ B.transpose(1, 2)).to(dev).contiguous()
    T0 = rnd(N, M, nv).to(dev)
    P1 = torch.diag(torch.rand(M, generator=g, dtype=torch.float64) + 0.5).to(dev)
    P0 = rnd(M, nv).to(dev)
    phi = rnd(N, M).to(dev)
    anc = torch.randint(0, N, (N,), generator=g).to(dev).to(torch.int32)
    idx = anc.long()
    sol = ops.mniw_solve(P0, P1, T0, T1, scale=0.97, anc=anc, phi=phi, keep_factor=True)
    eta1, eta0 = P1 + 0.97 * T1[idx], P0 + 0.97 * T0[idx]
    L = torch.linalg.cholesky(eta1)
    v = torch.linalg.solve_triangular(L, phi.unsqueeze(-1), upper=False)
    W = torch.linalg.solve_triangular(L, eta0, upper=False)
    ref = {"m": (W.transpose(1, 2) @ v).squeeze(-1), "q": W.transpose(1, 2) @ W, "c": (v * v).sum((1, 2)),
           "logdet": 2 * torch.log(torch.diagonal(L, dim1=1, dim2=2)).sum(1)}
    for k in ref:
        assert sol[k].shape == ref[k].shape, k
        err = (sol[k] - ref[k]).abs().max().item() / max(1.0, ref[k].abs().max().item())
        assert err < 1e-10, (k, err)
    assert torch.equal(sol["q"], sol["q"].transpose(1, 2))
    # the children reuse the factor of their ancestor
    phi2 = rnd(N, M).to(dev)
    anc2 = torch.randint(0, N, (N,), generator=g).to(dev).to(torch.int32)
    tri = ops.mniw_trisolve(sol, anc2, phi2)
    i2 = anc2.long()
    v2 = torch.linalg.solve_triangular(L[i2], phi2.unsqueeze(-1), upper=False)
    for k, r in (("m", (W[i2].transpose(1, 2) @ v2).squeeze(-1)), ("c", (v2 * v2).sum((1, 2)))):
        assert (tri[k] - r).abs().max().item() / max(1.0, r.abs().max().item()) < 1e-10, k
    # statistics: T_out = s T_in[a] + (phi xi^T, phi phi^T, xi xi^T, 1)
    xi = rnd(N, nv).to(dev)
    T2, T3 = rnd(N, nv, nv).to(dev), torch.rand(N, generator=g, dtype=torch.float64).to(dev)
    new = ops.stats_gather_update(0.9, anc, (T0, T1, T2, T3), phi, xi)
    want = (0.9 * T0[idx] + phi[:, :, None] * xi[:, None, :], 0.9 * T1[idx] + phi[:, :, None] * phi[:, None, :],
            0.9 * T2[idx] + xi[:, :, None] * xi[:, None, :], 0.9 * T3[idx] + 1.0)
    for a_, b_ in zip(new, want):
        assert a_.shape == b_.shape and (a_ - b_).abs().max().item() <= 1e-13 * max(1.0, b_.abs().max().item())
    w = torch.softmax(rnd(N).to(dev), 0)
    S = ops.weighted_stats(w, new)
    for s_, t_ in zip(S, new):
        r_ = torch.einsum("n...,n->...", t_, w)
        assert s_.shape == r_.shape and (s_ - r_).abs().max().item() <= 1e-12 * max(1.0, r_.abs().max().item())


def test_algorithm1_with_a_two_component_interface_variable():
    """n = 2 (not a configuration of the reference, whose interface variables are scalar; its formulas are general in n): the
    oscillator's force learnt as spring + damper components of one latent function -- against the NumPy restatement, eager and
    graph-replayed."""
    N = 150
    pb = experiments.smo_two_component_marginal(T=10)
    ref = marginal_oracle(pb, N)(CanonRand(SEED, N))
    for use_graph in (False, True):
        got = _device_alg(pb, N)(SEED, use_graph=use_graph)
        assert np.array_equal(got[4].cpu().numpy(), ref[4]), "ancestor_trace"
        _close(got[0], ref[0], "state_trace")
        _close(got[1][0], ref[1][0], "int_var_trace")
        assert got[1][0].shape == (pb.T, N, 2)
        for j in range(4):
            _close(got[2][0][j], ref[2][0][j], f"suff_stats_trace[{j}]")
            _close(got[5][0][j], ref[5][0][j], f"suff_stats[{j}]")
            assert tuple(got[5][0][j].shape) == np.shape(ref[5][0][j])
        _close(got[3], ref[3], "weights_trace")
        _close(got[7], ref[7], "log_likelihood", tol=1e-7)


def test_algorithm3_and_algorithm2_with_a_two_component_interface_variable():
    """The conditional filter (general log base measure: n M, n log det eta1, multigammaln(nu / 2, n), log det Psi -- BI:111-124) and
    the Particle-Gibbs chain over it, n = 2, against the NumPy restatement."""
    _algorithm3_case(experiments.smo_two_component_marginal(T=8), 120)
    _algorithm2_case(experiments.smo_two_component_marginal(T=7))


@pytest.mark.parametrize("N,M", [(300, 41), (257, 62), (70, 14), (64, 1)])
def test_mniw_wide_kernels_match_the_row_per_lane_kernel_bit_for_bit(N, M):
    """PGAS_OPT_MNIW_VALU = 2 runs the wide kernels at any M: same operations in the same order as the column-by-column kernel (= 1)."""
    ops = _ops(N)
    g = torch.Generator(device="cpu").manual_seed(11)
    dev = ops.device
    B = torch.randn(N, M, 3, generator=g, dtype=torch.float64)
    T1 = (B @ B.transpose(1, 2)).to(dev).contiguous()
    T0 = torch.randn(N, M, generator=g, dtype=torch.float64).to(dev)
    P1 = torch.diag(torch.rand(M, generator=g, dtype=torch.float64) + 0.5).to(dev)
    P0 = torch.randn(M, generator=g, dtype=torch.float64).to(dev)
    phi = torch.randn(N, M, generator=g, dtype=torch.float64).to(dev)
    anc = torch.randint(0, N, (N,), generator=g).to(dev).to(torch.int32)
    out = {}
    try:
        for knob in (1, 2):
            ops.eng.set_option(6, knob)
            full = ops.mniw_solve(P0, P1, T0, T1, scale=0.999, anc=anc, phi=phi, keep_factor=True)
            tri = ops.mniw_trisolve(full, anc, phi)
            out[knob] = {k: full[k].clone() for k in ("m", "c", "q", "logdet")} | {"tm": tri["m"].clone(), "tc": tri["c"].clone()}
    finally:
        ops.eng.set_option(6, 0)
    for k in out[1]:
        assert torch.equal(out[1][k], out[2][k]), k


@pytest.mark.parametrize("valu", [0, 1])
@pytest.mark.parametrize("N,M", [(300, 41), (1000, 20), (257, 62), (500, 46), (300, 30), (70, 14), (64, 1), (130, 5)])
def test_mniw_solve_against_torch(N, M, valu):
    """Both factorisation kernels: the MFMA-blocked default and the column-by-column VALU one (PGAS_OPT_MNIW_VALU)."""
    ops = _ops(N)
    ops.eng.set_option(6, valu)
    try:
        _check_mniw_solve(ops, N, M)
    finally:
        ops.eng.set_option(6, 0)


def _check_mniw_solve(ops, N, M):
    g = torch.Generator(device="cpu").manual_seed(3)
    dev = ops.device
    B = torch.randn(N, M, 3, generator=g, dtype=torch.float64)
    T1 = (B @ B.transpose(1, 2)).to(dev).contiguous()
    T0 = torch.randn(N, M, generator=g, dtype=torch.float64).to(dev)
    P1 = torch.diag(torch.rand(M, generator=g, dtype=torch.float64) + 0.5).to(dev)
    P0 = torch.randn(M, generator=g, dtype=torch.float64).to(dev)
    phi = torch.randn(N, M, generator=g, dtype=torch.float64).to(dev)
    anc = torch.randint(0, N, (N,), generator=g).to(dev).to(torch.int32)
    R1 = (torch.randn(M, M, generator=g, dtype=torch.float64) * 0.1).to(dev)
    R1 = (R1 @ R1.T).contiguous()
    R0 = torch.randn(M, generator=g, dtype=torch.float64).to(dev)
    for scale, a, r0, r1 in ((1.0, None, None, None), (0.999, anc, None, None), (1.0, None, R0, R1)):
        sol = ops.mniw_solve(P0, P1, T0, T1, scale=scale, anc=a, R0=r0, R1=r1, phi=phi)
        idx = torch.arange(N, device=dev) if a is None else a.long()
        eta1 = P1 + scale * T1[idx] + (0 if r1 is None else r1)
        eta0 = (P0 + scale * T0[idx] + (0 if r0 is None else r0)).unsqueeze(-1)
        L = torch.linalg.cholesky(eta1)
        v = torch.linalg.solve_triangular(L, phi.unsqueeze(-1), upper=False)
        w = torch.linalg.solve_triangular(L, eta0, upper=False)
        ref = {"m": (w * v).sum((1, 2)), "c": (v * v).sum((1, 2)), "q": (w * w).sum((1, 2)), "logdet": 2 * torch.log(torch.diagonal(L, dim1=1, dim2=2)).sum(1)}
        for k in ref:
            err = (sol[k] - ref[k]).abs().max().item() / max(1.0, ref[k].abs().max().item())
            assert err < 1e-10, (k, err)
    # stored factor + triangular solve = the full solve (the full solve accumulates m, c as a Schur complement, the triangular
    # solve as dot products: agreement to rounding, not bit for bit)
    full = ops.mniw_solve(P0, P1, T0, T1, scale=0.999, phi=phi, keep_factor=True)
    phi2 = torch.randn(N, M, generator=g, dtype=torch.float64).to(dev)
    tri = ops.mniw_trisolve(full, anc, phi2)
    again = ops.mniw_solve(P0, P1, T0, T1, scale=0.999, anc=anc, phi=phi2, want=("m", "c"))
    for k in ("m", "c"):
        assert (tri[k] - again[k]).abs().max().item() <= 1e-12 * max(1.0, again[k].abs().max().item())
    # a matrix that is not positive definite is reported, not silently processed
    bad = T1.clone()
    bad[5] = -torch.eye(M, dtype=torch.float64, device=dev) * 10
    ops.check()
    ops.mniw_solve(P0, P1, T0, bad, phi=phi)
    with pytest.raises(Exception, match="positive definite"):
        ops.check()
    ops.check()   # the counter is cleared by the failed check


def test_stats_gather_update_exact():
    N, M = 700, 41
    ops = _ops(N)
    dev = ops.device
    g = torch.Generator(device="cpu").manual_seed(4)
    T = (torch.randn(N, M, generator=g, dtype=torch.float64).to(dev), torch.randn(N, M, M, generator=g, dtype=torch.float64).to(dev),
         torch.rand(N, generator=g, dtype=torch.float64).to(dev), torch.rand(N, generator=g, dtype=torch.float64).to(dev) * 10)
    phi = torch.randn(N, M, generator=g, dtype=torch.float64).to(dev)
    xi = torch.randn(N, generator=g, dtype=torch.float64).to(dev)
    anc = torch.sort(torch.randint(0, N, (N,), generator=g)).values.to(dev).to(torch.int32)
    for lam, a in ((0.999, anc), (1.0, None)):
        out = ops.stats_gather_update(lam, a, T, phi, xi)
        idx = torch.arange(N, device=dev) if a is None else a.long()
        ref = (lam * T[0][idx] + phi * xi[:, None], lam * T[1][idx] + phi[:, :, None] * phi[:, None, :], lam * T[2][idx] + xi * xi, lam * T[3][idx] + 1.0)
        for o_, r_ in zip(out, ref):
            assert torch.equal(o_, r_)


@pytest.mark.parametrize("N,M", [(700, 41), (5000, 20), (513, 1)])
def test_weighted_stats_against_torch(N, M):
    ops = _ops(N)
    dev = ops.device
    g = torch.Generator(device="cpu").manual_seed(5)
    T = (torch.randn(N, M, generator=g, dtype=torch.float64).to(dev), torch.randn(N, M, M, generator=g, dtype=torch.float64).to(dev),
         torch.rand(N, generator=g, dtype=torch.float64).to(dev), torch.rand(N, generator=g, dtype=torch.float64).to(dev) * 10)
    w = torch.softmax(torch.randn(N, generator=g, dtype=torch.float64), 0).to(dev)
    S = ops.weighted_stats(w, T)
    ref = (w @ T[0], (w @ T[1].reshape(N, -1)).reshape(M, M), w @ T[2], w @ T[3])
    for s_, r_ in zip(S, ref):
        assert (s_ - r_).abs().max().item() <= 1e-13 * max(1.0, r_.abs().max().item())


def _device_alg(pb, N, kind="Algorithm1"):
    ssm = pb.ssm(pgas_amd.StateSpaceModel, torch)
    args = dict(N_samples=N, observations=pb.observations, inputs=pb.inputs, SSM=ssm, init_state_mean=pb.init_state_mean,
                init_state_cov=pb.init_state_cov, init_int_var_mean=pb.init_int_var_mean, init_int_var_cov=pb.init_int_var_cov,
                GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn())
    if kind == "Algorithm1":
        return pgas_amd.Algorithm1(forgetting_factor=pb.forgetting_factor, **args)
    return pgas_amd.Algorithm3(**args)


def _close(gpu, ref, what, tol=1e-8):
    g = gpu.cpu().numpy().reshape(np.shape(ref))
    err = np.abs(g - ref).max() / max(1.0, np.abs(ref).max())
    assert err < tol, f"{what}: relative error {err:.3e}"


def _problem(name, T=8):
    return {"smo": experiments.smo_marginal, "toy": experiments.toy_marginal, "vehicle": experiments.vehicle_marginal,
            "emps": experiments.emps_marginal}[name](T=T)


@pytest.mark.parametrize("name,N", [("smo", 200), ("toy", 200), ("smo", 1500), ("vehicle", 300), ("emps", 300)])
def test_algorithm1_matches_restatement(name, N):
    pb = _problem(name)
    ref = marginal_oracle(pb, N)(CanonRand(SEED, N))
    got = _device_alg(pb, N)(SEED)
    assert np.array_equal(got[4].cpu().numpy(), ref[4]), "ancestor_trace"
    _close(got[0], ref[0], "state_trace")
    for i in range(len(pb.basis)):          # Vehicle carries two latent functions
        _close(got[1][i], ref[1][i], f"int_var_trace[{i}]")
        for j in range(4):
            _close(got[2][i][j], ref[2][i][j], f"suff_stats_trace[{i}][{j}]")
            _close(got[5][i][j], ref[5][i][j], f"suff_stats[{i}][{j}]")
    _close(got[3], ref[3], "weights_trace")
    _close(got[6], ref[6], "obs_trace")
    _close(got[7], ref[7], "log_likelihood", tol=1e-7)
    assert got[2][0][0].shape == (pb.T, pb.GP_prior[0][0].shape[0], 1) and got[5][0][2].shape == (N, 1, 1)   # the reference's shapes


def test_marginalised_filter_with_a_wide_basis():
    """M = 80 basis functions per latent function (the reference runs Vehicle with 20): the filter steps go through the
    two-rows-per-lane MNIW kernels and still match the NumPy restatement.  (Algorithm1 only: the literal restatement of the
    conditional filter's base measures takes log(det eta1) as the reference does, BI:119, which overflows at this size -- the
    device works with log det throughout.)"""
    N = 100
    pb = experiments.vehicle_marginal(T=6, M=80)
    ref = marginal_oracle(pb, N)(CanonRand(SEED, N))
    got = _device_alg(pb, N)(SEED)
    assert np.array_equal(got[4].cpu().numpy(), ref[4]), "ancestor_trace"
    _close(got[0], ref[0], "state_trace")
    for i in range(len(pb.basis)):
        _close(got[1][i], ref[1][i], f"int_var_trace[{i}]")
        for j in range(4):
            _close(got[5][i][j], ref[5][i][j], f"suff_stats[{i}][{j}]")
    _close(got[3], ref[3], "weights_trace")


def test_step_glue_kernels_against_torch():
    """pgas_m_hilbert_basis, pgas_m_rng_student_t_df and pgas_m_mniw_draw against the torch expressions they replace."""
    N = 900
    ops = _ops(N)
    dev = ops.device
    g = torch.Generator(device="cpu").manual_seed(4)
    for pb in (experiments.smo_marginal(T=4), experiments.emps_marginal(T=4), experiments.toy_marginal(T=4)):
        bmap = pb.basis[0]
        nx = pb.init_state_mean.shape[0]
        st = (torch.randn(N, nx, generator=g, dtype=torch.float64) * 0.4).to(dev)
        u = torch.as_tensor(np.asarray(pb.inputs, dtype=np.float64).reshape(pb.T, -1)[1], device=dev)
        want = bmap.batch(st, u)                       # torch expression (nothing bound)
        got = ops.hilbert_basis(bmap, st, u)
        assert got.shape == want.shape and (got - want).abs().max().item() <= 1e-14 * max(1.0, want.abs().max().item())
        ref = bmap.batch(st.cpu().numpy(), u.cpu().numpy())
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-13
    anc = torch.randint(0, N, (N,), generator=g).to(dev).to(torch.int32)
    T3 = (torch.rand(N, generator=g, dtype=torch.float64) * 40).to(dev)
    t1 = ops.student_t_df(SEED, 32, 5, anc, T3, 3.0, 0.999)
    t2 = ops.student_t(SEED, 32, 5, 3.0 + 0.999 * T3[anc.long()])
    assert torch.equal(t1, t2)
    m, c, q, T2 = [torch.rand(N, generator=g, dtype=torch.float64).to(dev) for _ in range(4)]
    T2 = T2 + 2.0
    got = ops.mniw_draw(0.999, anc, m, c, q, T2, T3, 1.5, 3.0, t1)
    ai = anc.long()
    want = m + torch.sqrt((1.5 + 0.999 * T2[ai] - q[ai]) / (3.0 + 0.999 * T3[ai])) * t1 * torch.sqrt(c + 1.0)
    assert (got - want).abs().max().item() <= 1e-14 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("name", ["smo", "vehicle", "emps", "toy", "smo2"])
def test_traced_model_programs_match_the_torch_callables(name):
    """pgas_amd.SymbolicStateSpaceModel: transition / output / draw_state / log_likelihood as one launch each (pgas_m_expr_eval) against
    the torch callables they were traced from, with and without an ancestor gather."""
    pb = experiments.smo_two_component_marginal(T=6) if name == "smo2" else _problem(name, T=6)
    N = 700
    ops = _ops(N)
    dev = ops.device
    plain = pb.ssm(pgas_amd.StateSpaceModel, torch)
    sym = pb.ssm_symbolic(pgas_amd.SymbolicStateSpaceModel)
    sym.bind(ops)
    g = torch.Generator(device="cpu").manual_seed(2)
    nx = pb.init_state_mean.shape[0]
    st = (torch.randn(N, nx, generator=g, dtype=torch.float64) * 0.3).to(dev)
    u = torch.as_tensor(np.asarray(pb.inputs, dtype=np.float64).reshape(pb.T, -1)[2], device=dev)
    y = torch.as_tensor(np.asarray(pb.observations, dtype=np.float64).reshape(pb.T, -1)[2], device=dev)
    ivs = [(torch.randn(N, np.asarray(m).reshape(-1).shape[0], generator=g, dtype=torch.float64) * 0.5).to(dev) for m in pb.init_int_var_mean]
    z = torch.randn(N, nx, generator=g, dtype=torch.float64).to(dev)
    anc = torch.randint(0, N, (N,), generator=g).to(dev).to(torch.int32)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")   # a model that falls back to torch would say so
        pairs = [(sym.transition_mdl(st, u, *ivs), plain.transition_mdl(st, u, *ivs)),
                 (sym.output_mdl(st, u, *ivs).reshape(N, -1), plain.output_mdl(st, u, *ivs).reshape(N, -1)),
                 (sym.draw_state(z, st, u, *ivs), plain.draw_state(z, st, u, *ivs)),
                 (sym.draw_state_gather(z, st, u, anc, *ivs), plain.draw_state(z, st[anc.long()], u, *[v[anc.long()] for v in ivs])),
                 (sym.log_likelihood(y, st, u, *ivs), plain.log_likelihood(y, st, u, *ivs))]
    for k, (a_, b_) in enumerate(pairs):
        assert a_.shape == b_.shape, k
        assert (a_ - b_).abs().max().item() <= 1e-12 * max(1.0, b_.abs().max().item()), k
    assert all(p is not None for p in sym._progs.values()) and len(sym._progs) == 2, "both callables were traced"


@pytest.mark.parametrize("name", ["smo", "vehicle"])
def test_algorithm1_with_traced_model_matches_restatement(name):
    """The filter with the model's callables running as traced programs: same results as with the torch callables (and the restatement)."""
    N = 200
    pb = _problem(name)
    ref = marginal_oracle(pb, N)(CanonRand(SEED, N))
    args = dict(N_samples=N, observations=pb.observations, inputs=pb.inputs, SSM=pb.ssm_symbolic(pgas_amd.SymbolicStateSpaceModel),
                init_state_mean=pb.init_state_mean, init_state_cov=pb.init_state_cov, init_int_var_mean=pb.init_int_var_mean,
                init_int_var_cov=pb.init_int_var_cov, GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn())
    got = pgas_amd.Algorithm1(forgetting_factor=pb.forgetting_factor, **args)(SEED)
    assert np.array_equal(got[4].cpu().numpy(), ref[4]), "ancestor_trace"
    _close(got[0], ref[0], "state_trace")
    for i in range(len(pb.basis)):
        _close(got[1][i], ref[1][i], f"int_var_trace[{i}]")
    _close(got[3], ref[3], "weights_trace")
    _close(got[7], ref[7], "log_likelihood", tol=1e-7)


def _flat(out):
    st, iv, sst, w, anc, stats, obs, ll = out
    return [st, w, anc, obs, ll] + list(iv) + [t for s in sst for t in s] + [t for s in stats for t in s]


@pytest.mark.parametrize("name", ["smo", "toy", "vehicle", "emps"])
def test_algorithm1_graph_replay_equals_eager_loop(name):
    """Algorithm1.__call__ captures one filter step in a HIP graph and replays it for t = 3 .. T-1 (time index, Philox counters and
    trace rows addressed through device memory): every output must equal the eager loop's bit for bit."""
    pb = _problem(name, T=12)
    eager = _device_alg(pb, 200)(SEED, use_graph=False)
    graphed = _device_alg(pb, 200)(SEED, use_graph=True)
    for k, (a, b) in enumerate(zip(_flat(eager), _flat(graphed))):
        assert a.shape == b.shape and torch.equal(a, b), f"output {k} differs between the eager loop and the graph replay"


@pytest.mark.parametrize("name,N", [("smo", 150), ("toy", 150), ("vehicle", 200)])
def test_algorithm3_matches_restatement(name, N):
    _algorithm3_case(_problem(name), N)


def _algorithm3_case(pb, N):
    oracle = marginal_oracle(pb, N, "Algorithm3")
    ref_x, ref_iv = pb.X_true, list(pb.int_var_true)
    ref_stats = mo.trajectory_stats(oracle, ref_x, ref_iv)
    traj, ivt, tr = oracle(CanonRand(SEED, N), ref_x, ref_iv, ref_stats)
    alg = _device_alg(pb, N, "Algorithm3")
    gt, gi, gtr = alg(SEED, ref_x, ref_iv, ref_stats, return_traces=True)
    assert np.array_equal(gtr["ancestor_trace"].cpu().numpy(), tr["ancestor_trace"]) and gtr["idx"] == tr["idx"]
    _close(gtr["state_trace"], tr["state_trace"], "state_trace")
    _close(gtr["log_weights"], tr["log_weights"], "final log-weights", tol=1e-7)
    _close(gt, traj, "state trajectory")
    for i in range(len(pb.basis)):
        _close(gi[i], ivt[i], f"interface-variable trajectory {i}")


@pytest.mark.parametrize("name", ["smo", "toy", "vehicle"])
def test_algorithm3_graph_replay_equals_eager_loop(name):
    """The conditional filter's loop (src/Algorithm3.py:251-290) captured once and replayed: traces, trajectory and final index equal
    the eager loop's bit for bit (reference rows, reference statistics and both uniforms are addressed on the device)."""
    pb = _problem(name, T=12)
    ref_x, ref_iv = pb.X_true, list(pb.int_var_true)
    ref_stats = mo.trajectory_stats(marginal_oracle(pb, 200, "Algorithm3"), ref_x, ref_iv)
    outs = []
    for mode in (False, True):
        gt, gi, gtr = _device_alg(pb, 200, "Algorithm3")(SEED, ref_x, ref_iv, ref_stats, return_traces=True, use_graph=mode)
        outs.append([gt, gtr["state_trace"], gtr["ancestor_trace"], gtr["log_weights"]] + list(gi) + [torch.tensor(gtr["idx"])])
    for k, (a, b) in enumerate(zip(*outs)):
        assert a.shape == b.shape and torch.equal(a.cpu(), b.cpu()), f"output {k} differs between the eager loop and the graph replay"


@pytest.mark.parametrize("name", ["smo", "vehicle"])
def test_algorithm2_matches_restatement(name):
    """Whole Particle-Gibbs chains (Algorithm2 over Algorithm3): the device mirror splits its key once per iteration
    (pgas_amd.random.split); the restatement is given providers on those same per-iteration seeds."""
    _algorithm2_case(_problem(name, T=7))


def _algorithm2_case(pb):
    from pgas_amd import random as prng

    N, K = 96, 4
    ssm_t, ssm_n = pb.ssm(pgas_amd.StateSpaceModel, torch), pb.ssm(mo.StateSpaceModel, np)
    common = dict(N_samples=N, N_iterations=K, observations=pb.observations, inputs=pb.inputs, init_state_mean=pb.init_state_mean,
                  init_state_cov=pb.init_state_cov, init_int_var_mean=pb.init_int_var_mean, init_int_var_cov=pb.init_int_var_cov,
                  GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn())
    got = pgas_amd.Algorithm2(SSM=ssm_t, **common)(SEED, pb.X_true, list(pb.int_var_true))

    def rands():
        key = prng.as_key(SEED)
        while True:
            key, key_step = prng.split(key, 2)
            yield CanonRand(key_step, N)

    ref = mo.Algorithm2(SSM=ssm_n, **common)(rands(), pb.X_true, list(pb.int_var_true))
    _close(got[0], ref[0], "state_trace", tol=1e-7)
    for i in range(len(pb.basis)):
        _close(got[1][i], ref[1][i], f"int_var_trace[{i}]", tol=1e-7)
        for j in range(4):
            _close(got[3][i][j], ref[3][i][j], f"suff_stats_trace[{i}][{j}]", tol=1e-7)
    _close(got[4], ref[4], "obs_trace", tol=1e-7)
    _close(got[5], ref[5], "log_likelihood", tol=1e-6)


def test_algorithm2_runs_and_returns_reference_shapes():
    pb = experiments.toy_marginal(T=12)
    N, K = 64, 4
    ssm = pb.ssm(pgas_amd.StateSpaceModel, torch)
    alg = pgas_amd.Algorithm2(N_samples=N, N_iterations=K, observations=pb.observations, inputs=pb.inputs, SSM=ssm, init_state_mean=pb.init_state_mean,
                              init_state_cov=pb.init_state_cov, init_int_var_mean=pb.init_int_var_mean, init_int_var_cov=pb.init_int_var_cov,
                              GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn())
    X, IV, W, SS, OBS, LL = alg(SEED, pb.X_true, [pb.int_var_true[0]])
    T = pb.T
    assert X.shape == (T, K, 1) and IV[0].shape == (T, K, 1) and W.shape == (T, K) and OBS.shape == (T, K, 1) and LL.shape == (T, K)
    assert SS[0][0].shape == (K, 40, 1) and SS[0][1].shape == (K, 40, 40) and SS[0][3].shape == (K,)
    assert torch.isfinite(X).all() and torch.isfinite(LL).all() and torch.all(SS[0][3] == T)
    assert torch.allclose(X[:, 0, 0], torch.as_tensor(pb.X_true[:, 0], device=X.device))


@pytest.mark.parametrize("name", ["smo", "toy", "vehicle"])
def test_device_matches_committed_vectors(name):
    """The device path against tests/golden/marginal_runs.json (vectors of the NumPy restatement, tools/make_golden.py)."""
    import json
    import os

    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "marginal_runs.json")))[name]
    pb = _problem(name, T=g["T"])
    N = g["N"]
    got = _device_alg(pb, N)(g["seed"])
    assert np.array_equal(got[4].cpu().numpy(), np.array(g["alg1_ancestors"]))
    _close(got[0][-1].reshape(-1), np.array(g["alg1_state_last"]), "state (last step)")
    _close(got[3][-1], np.array(g["alg1_weights_last"]), "weights (last step)", tol=1e-7)
    for i in range(len(pb.basis)):
        _close(got[1][i][-1].reshape(-1), np.array(g["alg1_int_var_last"][i]), f"int_var[{i}] (last step)")
    from oracle import marginal_numpy as mo2

    a3 = marginal_oracle(pb, N, "Algorithm3")
    ref_stats = mo2.trajectory_stats(a3, pb.X_true, list(pb.int_var_true))
    gt, gi, gtr = _device_alg(pb, N, "Algorithm3")(g["seed"], pb.X_true, list(pb.int_var_true), ref_stats, return_traces=True)
    assert np.array_equal(gtr["ancestor_trace"].cpu().numpy(), np.array(g["alg3_ancestors"])) and gtr["idx"] == g["alg3_idx"]
    _close(gt.reshape(-1), np.array(g["alg3_state_traj"]), "Algorithm3 state trajectory")
    for i in range(len(pb.basis)):
        _close(gi[i].reshape(-1), np.array(g["alg3_int_var_traj"][i]), f"Algorithm3 int_var trajectory {i}")


def test_a_model_that_cannot_be_captured_falls_back_or_says_why():
    """A model callable that synchronises with the host cannot be captured in a HIP graph.  The filter either warns and goes on launch by
    launch with the same results, or -- when the failed capture invalidated the stream -- raises an error that says so.  Run in a child
    process: a poisoned stream must not reach the other tests."""
    import os
    import subprocess
    import sys

    code = r"""
import sys, warnings
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from common import experiments, pgas_amd
pb = experiments.smo_marginal(T=8)
def make(sync):
    f, g = pb.model(torch)
    def f_sync(state, input, *iv):
        if sync:
            float(state[0, 0].item())          # host round trip: not capturable
        return f(state, input, *iv)
    ssm = pgas_amd.StateSpaceModel(pb.process_noise, pb.output_noise, f_sync, g)
    return pgas_amd.Algorithm1(N_samples=64, observations=pb.observations, inputs=pb.inputs, SSM=ssm, forgetting_factor=pb.forgetting_factor,
                               init_state_mean=pb.init_state_mean, init_state_cov=pb.init_state_cov, init_int_var_mean=pb.init_int_var_mean,
                               init_int_var_cov=pb.init_int_var_cov, GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn())
ref = make(False)(7, use_graph=False)[0].cpu().numpy()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    try:
        got = make(True)(7, use_graph=True)[0].cpu().numpy()
        print("WARNED" if any("could not be captured" in str(x.message) for x in w) else "SILENT", "EQUAL" if np.array_equal(got, ref) else "DIFFERENT")
    except RuntimeError as e:
        print("REFUSED" if "unusable" in str(e) else "OTHER " + str(e)[:200])
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    last = [ln for ln in out.stdout.splitlines() if ln.strip()][-1] if out.stdout.strip() else out.stderr[-400:]
    assert last in ("WARNED EQUAL", "REFUSED"), (last, out.stderr[-600:])
