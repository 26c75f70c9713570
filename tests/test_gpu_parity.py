"""GPU parity tests: the HIP engine (through the C ABI, via pgas_amd) against the canonical C oracle.

Bar: BIT-EXACT on every output (states, log-weights, ancestor indices, trajectory) -- the canonical
arithmetic of DESIGN.md section 4 makes fp64 results reproducible across host and device.
The oracle itself is pinned against the literal NumPy restatement of the reference in
tests/test_oracle_canon.py (tolerance 1e-12, indices equal away from CDF ties).
"""
import numpy as np
import pytest
import torch

from common import canon_model, experiments, pgas_amd

pytestmark = pytest.mark.gpu

SEED = 12345678


def _problems():
    return {
        "smo": lambda: experiments.smo_pgas(T=40),
        "toy": lambda: experiments.toy(T=40),
        "emps": lambda: experiments.emps_pgas(T=10),
        "emps27": lambda: experiments.emps_pgas(T=16, M=27),
        "veh": lambda: experiments.vehicle_pgas(T=12),           # ny = 2, nu = 2, M = 729 (BASELINE configs[2])
        "veh27": lambda: experiments.vehicle_pgas(T=300, M=27),
    }


def _setup(name, N):
    pb = _problems()[name]()
    A, S = experiments.initial_params(pb)
    cm = canon_model(pb, N)
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov,
                                             pb.likelihood_fcn, pb.basis_fcn)
    return pb, A, S, cm, csmc


def _eq(gpu, ref, what):
    g = gpu.cpu().numpy().reshape(np.shape(ref))
    assert np.array_equal(g, ref), f"{what}: {int((g != ref).sum())} of {g.size} entries differ, max |d| = {np.abs(g - ref).max():.3e}"


def test_library_loaded_is_in_tree():
    from pgas_amd import _lib

    _lib.load()
    assert _lib.LIB_PATH.endswith("libpgas_hip.so")
    maps = open("/proc/self/maps").read()
    assert "libpgas_hip.so" in maps, "native library not mapped into the test process"


@pytest.mark.parametrize("name,N", [("smo", 200), ("smo", 1024), ("smo", 1025), ("smo", 70000), ("toy", 777), ("emps", 1500), ("emps27", 3000), ("veh", 1300), ("veh27", 2500)])
def test_basis_init_step_bit_exact(name, N):
    pb, A, S, cm, csmc = _setup(name, N)
    eng = csmc.engine
    LS, LSinv, cS = cm.chol_parts(S)
    L0 = np.linalg.cholesky(pb.init_state_cov)
    rng = np.random.default_rng(5)
    xs = pb.X_true[rng.integers(0, pb.T, 257)] + 0.05 * rng.standard_normal((257, pb.nx))
    _eq(eng.basis_eval(xs, 2), cm.basis_eval(xs, 2), "basis_eval")
    x = cm.init_state(SEED, pb.init_state_mean, L0, pb.X_true[0])
    _eq(eng.init_state(SEED, pb.X_true[0]), x, "init_state")
    lw = None
    for t in (1, 2, 3):
        lwo, xo, ao, dbg = cm.step(t, SEED, x, lw, A, LS, LSinv, cS, pb.X_true[t], debug=True)
        if t == 1:
            eng.set_params(A, S)
            _eq(eng.aux_states(x, t), dbg["aux"], "aux_states")
        lwg, xg, ag = csmc.step(SEED, t, lw, x, A, S, pb.X_true[t])
        _eq(xg, xo, f"step {t} new_state")
        _eq(ag, ao, f"step {t} a_indices")
        _eq(lwg, lwo, f"step {t} new_log_weights")
        # conditional-SMC invariants
        assert np.array_equal(xg[-1].cpu().numpy(), pb.X_true[t].reshape(-1))
        assert np.all(np.diff(ag[:-1].cpu().numpy()) >= 0), "systematic resampling indices must be non-decreasing"
        lw, x = lwo, xo


@pytest.mark.parametrize("name,N,opts", [
    ("smo", 200, {}), ("smo", 4096, {}), ("smo", 5000, {}), ("toy", 1500, {}), ("emps", 2048, {}), ("veh", 2048, {}), ("veh27", 3000, {}), ("smo", 1 << 17, {}),
    ("smo", 5000, {7: 1}),            # PGAS_OPT_LOCAL_GROUPS: k_step<LOCAL>, every workgroup scans all groups itself
    ("smo", 70000, {7: 1, 1: 7}),     # ... with k_propagate launched in chunks of 7 time steps
    ("smo", 1 << 17, {7: 1}),         # ... two groups
    ("smo", 1 << 17, {9: 1}),         # PGAS_OPT_TAIL_GROUPS: group scans handed to the last-arriving workgroup inside k_step
    ("smo", 70000, {9: 1}),           # ... ragged last group
    ("toy", 1500, {1: 1}),            # PGAS_OPT_PROPAGATE_CHUNK = 1: one k_propagate launch per step
    ("smo", 5000, {8: 3, 11: 2}),     # PGAS_OPT_EVENT_STRIDE = 3, PGAS_OPT_MAX_LEAD = 2: k_propagate at most two event groups ahead of the chain
    ("smo", 5000, {3: 0}),            # PGAS_OPT_OVERLAP = 0: both pipelines on the caller's stream
    ("smo", 5000, {13: 1}), ("emps", 2048, {13: 1}), ("toy", 300, {13: 1}), ("smo", 70000, {13: 1, 1: 7}),   # PGAS_OPT_GRAPH = 1: the sweep captured in a HIP graph and replayed
    ("smo", 5000, {12: 1 << 18}),     # PGAS_OPT_TRACE_BLOCK_BYTES: traces in row blocks (2 state rows, 4 hand-off rows, 8 ancestor rows per block)
    ("smo", 70000, {12: 4 << 20, 1: 7}),   # ... with k_propagate chunks of 7 steps that straddle block boundaries (split launches)
    ("emps", 2048, {12: 1 << 17}),    # ... 3-D basis, 4 state rows per block
    # PGAS_OPT_MFMA_PROPAGATE = 1: the 3-D contraction's innermost sum on the f64 matrix cores (default: vector ALUs); odd segment counts, a one-particle last segment
    ("emps", 2048, {15: 1}), ("veh", 2048, {15: 1}), ("emps", 3000, {15: 1}), ("veh", 5000, {15: 1}), ("emps", 1025, {15: 1}), ("emps", 2048, {15: 1, 1: 5}), ("emps", 3000, {}),
    ("toy", 1500, {12: 16384}),       # ... one row per block
    # N <= 1024: one segment -- by default the whole sweep is one launch of one workgroup (k_sweep_small); 14: 0 = PGAS_OPT_SMALL_SWEEP off,
    # the general multi-launch path at the same sizes
    ("smo", 200, {14: 0}), ("smo", 1024, {14: 0}), ("toy", 300, {14: 0}), ("emps", 500, {14: 0}), ("veh", 640, {14: 0}), ("smo", 777, {14: 0, 7: 1}), ("toy", 1, {14: 0}),
    ("smo", 256, {}), ("smo", 257, {}), ("toy", 1024, {}), ("emps27", 1000, {}), ("smo", 513, {13: 1, 14: 0}),
    ("emps", 250, {}), ("veh", 130, {}), ("veh27", 256, {}), ("toy", 255, {}), ("smo", 2, {}), ("smo", 63, {}),
    # 14: 2 = the one-workgroup kernel (k_sweep_small) where the default is the two-workgroup pipeline (k_sweep_duo)
    ("smo", 200, {14: 2}), ("toy", 1, {14: 2}), ("emps", 250, {14: 2}), ("smo", 1024, {14: 2}), ("veh27", 600, {14: 2}),
])
def test_sweep_bit_exact(name, N, opts):
    pb, A, S, cm, csmc = _setup(name, N)
    for k, v in opts.items():
        csmc.engine.set_option(k, v)
    LS, LSinv, cS = cm.chol_parts(S)
    traj = csmc(SEED, pb.X_true, A, S)
    trajo, Xo, ANCo, lwo = cm.sweep(SEED, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov))
    X, ANC, LW, _ = csmc.engine.traces()
    _eq(X, Xo, "state_trace")
    _eq(ANC[: pb.T - 1], ANCo, "ancestor_trace")
    _eq(LW, lwo, "log_weights_trace[-1]")
    _eq(traj, trajo.reshape(traj.shape), "trajectory")
    assert csmc.engine.launch_info()["small"] == (N <= 1024 and opts.get(14, 1) != 0), "which sweep ran"
    assert csmc.engine.launch_info()["mfma"] == (opts.get(15, 0) == 1 and name in ("emps", "veh") and N > 1024), "which k_propagate ran"
    # the trajectory is a path through the trace (src/Filtering.py:40-55)
    b = csmc.engine.last_final_index()
    Xn, An = X.cpu().numpy(), ANC.cpu().numpy()
    for t in range(pb.T - 1, -1, -1):
        assert np.array_equal(Xn[t, b], trajo[t].reshape(-1))
        if t:
            b = An[t - 1, b]


@pytest.mark.parametrize("name,N", [("smo", 3000), ("emps", 1500), ("veh", 1300)])
def test_generic_propagate_variant_bit_exact(name, N, monkeypatch):
    """The reference's model shapes (7 x 7 and 9 x 9 x 9 frequency grids, inputs in natural order) run k_propagate instantiations
    specialised at compile time; PGAS_NO_FAST_VARIANT=1 forces the generic instantiation the other shapes take.  Both must give
    the oracle's sweep bit for bit."""
    monkeypatch.setenv("PGAS_NO_FAST_VARIANT", "1")
    pb, A, S, cm, csmc = _setup(name, N)
    LS, LSinv, cS = cm.chol_parts(S)
    traj = csmc(SEED, pb.X_true, A, S)
    trajo, Xo, ANCo, lwo = cm.sweep(SEED, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov))
    X, ANC, LW, _ = csmc.engine.traces()
    _eq(X, Xo, "state_trace")
    _eq(ANC[: pb.T - 1], ANCo, "ancestor_trace")
    _eq(traj, trajo.reshape(traj.shape), "trajectory")
    info = csmc.engine.launch_info()
    assert info["JP"] in (8, 12), info   # padded grid extent of the generic instantiations (the fast ones use the exact 7 / 9)


def test_graph_replay_follows_seed_parameters_and_reference():
    """The captured sweep (PGAS_OPT_GRAPH) is replayed with everything that changes between sweeps -- seed, uniforms,
    (A, S), reference trajectory -- read from device memory at execution time: three replays with different inputs, then the first
    inputs again, each bit-identical to the oracle (and the first and last to each other)."""
    pb = experiments.smo_pgas(T=20)
    N = 6000
    A, S = experiments.initial_params(pb)
    cm = canon_model(pb, N)
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn)
    csmc.engine.set_option(13, 1)   # PGAS_OPT_GRAPH (off by default: slower than enqueueing on this runtime)
    L0 = np.linalg.cholesky(pb.init_state_cov)
    rng = np.random.default_rng(4)
    cases = [(SEED, A, S, pb.X_true),
             (SEED + 1, A * 0.97, S * 1.3, pb.X_true + 1e-3 * rng.standard_normal(pb.X_true.shape)),
             (SEED + 2, A + 1e-3 * rng.standard_normal(A.shape), S, pb.X_true),
             (SEED, A, S, pb.X_true)]
    got = []
    for seed, a, s_, ref in cases:
        LS, LSinv, cS = cm.chol_parts(s_)
        traj = csmc(seed, ref, a, s_)
        trajo, Xo, ANCo, lwo = cm.sweep(seed, ref, a, LS, LSinv, cS, pb.init_state_mean, L0)
        X, ANC, LW, _ = csmc.engine.traces()
        _eq(X, Xo, "state_trace")
        _eq(ANC[: pb.T - 1], ANCo, "ancestor_trace")
        _eq(traj, trajo.reshape(traj.shape), "trajectory")
        assert csmc.engine.launch_info()["graph"]
        got.append(traj.cpu().numpy())
    assert np.array_equal(got[0], got[3]) and not np.array_equal(got[0], got[1])


def test_sweep_degenerate_weights():
    """One observation far from every particle: a handful of particles carry all the weight."""
    pb = experiments.smo_pgas(T=12)
    pb.observations = pb.observations.copy()
    pb.observations[5] += 0.5  # ~16 sigma of R = 1e-3
    N = 6000
    A, S = experiments.initial_params(pb)
    cm = canon_model(pb, N)
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov,
                                             pb.likelihood_fcn, pb.basis_fcn)
    LS, LSinv, cS = cm.chol_parts(S)
    traj = csmc(SEED, pb.X_true, A, S)
    trajo, Xo, ANCo, lwo = cm.sweep(SEED, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov))
    X, ANC, LW, _ = csmc.engine.traces()
    _eq(ANC[: pb.T - 1], ANCo, "ancestor_trace")
    _eq(traj, trajo, "trajectory")
    assert len(np.unique(ANCo[4])) < N // 4  # the step really was degenerate


@pytest.mark.parametrize("N", [3000, 300])
def test_logw_trace_option(N):
    pb = experiments.smo_pgas(T=8)
    A, S = experiments.initial_params(pb)
    cm = canon_model(pb, N)
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov,
                                             pb.likelihood_fcn, pb.basis_fcn, keep_logw_trace=True)
    LS, LSinv, cS = cm.chol_parts(S)
    csmc(SEED, pb.X_true, A, S)
    _, _, _, LT = csmc.engine.traces()
    x = cm.init_state(SEED, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov), pb.X_true[0])
    lw = None
    assert float(LT[0].abs().max()) == 0.0
    for t in range(1, pb.T):
        lw, x, _ = cm.step(t, SEED, x, lw, A, LS, LSinv, cS, pb.X_true[t])
        _eq(LT[t], lw, f"log_weights_trace[{t}]")


def test_full_size_properties():
    """BASELINE size N = 2^20 (T shortened to 18): size-independent properties of the whole sweep, and EVERY one of its 17 steps checked
    bit for bit against the canonical C oracle (teacher-forced with the device's own previous states and log-weights, which are
    themselves checked one step earlier): this is the k_step path of the bench -- 1024 segments, 16 groups, staged windows -- at full size."""
    T = 18
    pb = experiments.smo_pgas(T=T)
    N = 1 << 20
    A, S = experiments.initial_params(pb)
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov,
                                             pb.likelihood_fcn, pb.basis_fcn, keep_logw_trace=True)
    traj = csmc(SEED, pb.X_true, A, S)
    assert not csmc.engine.launch_info()["local_groups"], "the bench's default path: k_groups between the steps"
    X, ANC, LW, LT = csmc.engine.traces()
    a = ANC[: pb.T - 1]
    assert int(a.min()) >= 0 and int(a.max()) < N
    assert bool((a[:, 1:-1] >= a[:, :-2]).all()), "resampled indices must be sorted"
    assert torch.equal(X[:, -1, :], torch.as_tensor(pb.X_true, device=X.device)), "conditioned particle must follow the reference"
    assert bool(torch.isfinite(LW).all())
    cm = canon_model(pb, N)
    LS, LSinv, cS = cm.chol_parts(S)
    x0 = cm.init_state(SEED, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov), pb.X_true[0])
    _eq(X[0], x0, "x_trace[0] at N=2^20")
    for t in range(1, T):
        xp = X[t - 1].cpu().numpy()
        lwp = None if t == 1 else LT[t - 1].cpu().numpy()
        lwo, xo, ao, dbg = cm.step(t, SEED, xp, lwp, A, LS, LSinv, cS, pb.X_true[t], debug=True)
        _eq(X[t], xo, f"x_trace[{t}] at N=2^20")
        _eq(ANC[t - 1], ao, f"anc_trace[{t - 1}] at N=2^20")
        _eq(LT[t], lwo, f"log_weights_trace[{t}] at N=2^20")
        if t in (1, T - 1):
            # offspring counts of systematic resampling are within +-1 of N w
            w = np.exp(dbg["lw1"] - dbg["lw1"].max())
            w /= w.sum()
            counts = np.bincount(ao[:-1], minlength=N)
            assert np.all(np.abs(counts - N * w) <= 2.0)
    assert traj.shape == (pb.T, 2)
    b = csmc.engine.last_final_index()
    Xn, An = X.cpu().numpy(), ANC.cpu().numpy()
    for t in range(pb.T - 1, -1, -1):
        assert np.array_equal(Xn[t, b], traj[t].cpu().numpy())
        if t:
            b = An[t - 1, b]


@pytest.mark.parametrize("name,steps", [("smo", (500, 1000, 1500, 1998, 1999)), ("emps", (1200, 1999))])
def test_full_size_late_time_parity(name, steps):
    """The regime bench.py times: N = 2^20, T = 2000, weights after hundreds of resampling generations.  Teacher-forced oracle steps
    late in the sweep (inputs = the device's own previous states and log-weights), the final index (src/PGAS.py:224-225) and the whole
    back-trace (src/Filtering.py:40-55) chased independently through the device's traces -- all bit for bit."""
    T, N = 2000, 1 << 20
    pb = experiments.smo_pgas(T=T) if name == "smo" else experiments.emps_pgas(T=T)
    A, S = experiments.initial_params(pb)
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov,
                                             pb.likelihood_fcn, pb.basis_fcn, keep_logw_trace=True)
    traj = csmc(SEED, pb.X_true, A, S)
    X, ANC, LW, LT = csmc.engine.traces()
    assert torch.equal(X[:, -1, :], torch.as_tensor(pb.X_true, device=X.device)), "conditioned particle must follow the reference"
    assert bool(torch.isfinite(LT).all())
    cm = canon_model(pb, N)
    LS, LSinv, cS = cm.chol_parts(S)
    for t in steps:
        lwo, xo, ao = cm.step(t, SEED, X[t - 1].cpu().numpy(), LT[t - 1].cpu().numpy(), A, LS, LSinv, cS, pb.X_true[t])
        _eq(X[t], xo, f"{name}: x_trace[{t}] at N=2^20")
        _eq(ANC[t - 1], ao, f"{name}: anc_trace[{t - 1}] at N=2^20")
        _eq(LT[t], lwo, f"{name}: log_weights_trace[{t}] at N=2^20")
        assert len(np.unique(ao)) < N, "late steps do resample"
    _eq(LW, LT[T - 1].cpu().numpy(), "logw_last == log_weights_trace[T-1]")
    b = csmc.engine.last_final_index()
    assert b == cm.final_index(SEED, LW.cpu().numpy()), "final index"
    # the ancestral path of that index, chased here element by element (not with the library's kernel)
    bt = torch.tensor([b], device=X.device, dtype=torch.int64)
    path = torch.empty_like(traj)
    for t in range(T - 1, -1, -1):
        path[t] = X[t].index_select(0, bt)[0]
        if t:
            bt = ANC[t - 1].index_select(0, bt).to(torch.int64)
    assert torch.equal(path, traj), "trajectory != ancestral path of the final index"


@pytest.mark.parametrize("name", ["emps", "veh"])
def test_full_size_properties_m729(name):
    """BASELINE configs[2] / configs[4] at their full size (N = 2^20, M = 729, 3-D basis; T shortened): size-independent properties
    of the sweep and the first two steps against the canonical oracle bit for bit."""
    T = 4
    pb = experiments.emps_pgas(T=T) if name == "emps" else experiments.vehicle_pgas(T=T)
    N = 1 << 20
    A, S = experiments.initial_params(pb)
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov,
                                             pb.likelihood_fcn, pb.basis_fcn, keep_logw_trace=True)
    traj = csmc(SEED, pb.X_true, A, S)
    X, ANC, LW, LT = csmc.engine.traces()
    a = ANC[: T - 1]
    assert int(a.min()) >= 0 and int(a.max()) < N
    assert bool((a[:, 1:-1] >= a[:, :-2]).all()), "resampled indices must be sorted"
    assert torch.equal(X[:, -1, :], torch.as_tensor(pb.X_true, device=X.device)), "conditioned particle must follow the reference"
    assert bool(torch.isfinite(LW).all()) and traj.shape == (T, 2)
    cm = canon_model(pb, N)
    LS, LSinv, cS = cm.chol_parts(S)
    for t in (1, 2):
        lwo, xo, ao = cm.step(t, SEED, X[t - 1].cpu().numpy(), None if t == 1 else LT[t - 1].cpu().numpy(), A, LS, LSinv, cS, pb.X_true[t])
        _eq(X[t], xo, f"{name}: x_trace[{t}] at N=2^20")
        _eq(ANC[t - 1], ao, f"{name}: anc_trace[{t - 1}] at N=2^20")
        _eq(LT[t], lwo, f"{name}: log_weights_trace[{t}] at N=2^20")


def test_two_million_particles_group_path():
    """N = 2^21 on one device (2048 segments, 32 groups: the window of a workgroup no longer covers the device).
    Three steps against the oracle, bit for bit."""
    T = 4
    pb = experiments.smo_pgas(T=T)
    N = 1 << 21
    A, S = experiments.initial_params(pb)
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov,
                                             pb.likelihood_fcn, pb.basis_fcn, keep_logw_trace=True)
    csmc(SEED, pb.X_true, A, S)
    assert not csmc.engine.launch_info()["local_groups"]
    X, ANC, LW, LT = csmc.engine.traces()
    cm = canon_model(pb, N)
    LS, LSinv, cS = cm.chol_parts(S)
    for t in (1, 2, 3):
        lwo, xo, ao = cm.step(t, SEED, X[t - 1].cpu().numpy(), None if t == 1 else LT[t - 1].cpu().numpy(), A, LS, LSinv, cS, pb.X_true[t])
        _eq(X[t], xo, f"x_trace[{t}] at N=2^21")
        _eq(ANC[t - 1], ao, f"anc_trace[{t - 1}] at N=2^21")
        _eq(LT[t], lwo, f"log_weights_trace[{t}] at N=2^21")


def test_error_reporting():
    pb = experiments.smo_pgas(T=5)
    from pgas_amd._lib import PgasError

    csmc = pgas_amd.condSequentialMonteCarlo(64, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov,
                                             pb.likelihood_fcn, pb.basis_fcn)
    with pytest.raises(PgasError, match="set_params"):
        csmc.engine.sweep(1, pb.X_true)
    with pytest.raises(TypeError):
        pgas_amd.condSequentialMonteCarlo(64, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov,
                                          lambda o, s, i: 0.0, pb.basis_fcn)


@pytest.mark.parametrize("name,N,world,blk", [("smo", 4096, 2, None), ("smo", 8192, 4, 1 << 17), ("smo", 65536, 8, None), ("toy", 2048, 2, 8192), ("emps", 2048, 2, None),
                                               ("veh27", 4096, 2, 1 << 16),
                                               ("emps", 8192, 8, None), ("veh", 8192, 8, 1 << 16)])   # BASELINE configs[4]'s 8-way placement (M = 729), small N
def test_sharded_sweep_bit_exact_and_independent_of_world(name, N, world, blk):
    """Particle-sharded sweep (several shards emulated in one process on one device): the trajectory and the traces are the
    single-device / oracle ones bit for bit, whatever the number of shards."""
    from pgas_amd import sharded

    pb = _problems()[name]()
    A, S = experiments.initial_params(pb)
    cm = canon_model(pb, N)
    LS, LSinv, cS = cm.chol_parts(S)
    trajo, Xo, ANCo, lwo = cm.sweep(SEED, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov))
    # blk: traces in small row blocks (the default is 1 GiB: one block at these sizes), so that peers' rows, the chunked propagation
    # and the cross-rank ancestor chase all cross block boundaries
    grp = sharded.make_local_group(world, N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn,
                                   trace_block_bytes=blk)
    if blk:
        assert grp.shards[0].nblk[5] > 1, "the state trace should span several blocks in this case"
    trajs = sharded.sharded_sweep(grp, SEED, pb.X_true, A, S, propagate_chunk=5)
    Nl = N // world
    for r, (s, tr) in enumerate(zip(grp.shards, trajs)):
        _eq(tr, trajo.reshape(tr.shape), f"trajectory on rank {r}")
        X, ANC, LW, _ = s.eng.traces()
        _eq(X, Xo[:, r * Nl:(r + 1) * Nl], f"state_trace shard {r}")
        _eq(ANC[: pb.T - 1], ANCo[:, r * Nl:(r + 1) * Nl], f"ancestor_trace shard {r}")
        _eq(LW, lwo[r * Nl:(r + 1) * Nl], f"log_weights shard {r}")


def test_eight_shards_five_million_particles():
    """The geometry of BASELINE config 4 beyond what one window holds: 8 shards x 640 segments = 5120 segments = 80 groups (the top
    level scans two blocks of groups, every workgroup's window is a small part of the device, most peers are remote).  T = 3,
    traces and trajectory against the oracle bit for bit."""
    from pgas_amd import sharded

    world, Nl = 8, 640 * 1024
    N = world * Nl
    pb = experiments.smo_pgas(T=3)
    A, S = experiments.initial_params(pb)
    cm = canon_model(pb, N)
    LS, LSinv, cS = cm.chol_parts(S)
    trajo, Xo, ANCo, lwo = cm.sweep(SEED, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov))
    grp = sharded.make_local_group(world, N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn)
    trajs = sharded.sharded_sweep(grp, SEED, pb.X_true, A, S)
    for r, (s, tr) in enumerate(zip(grp.shards, trajs)):
        _eq(tr, trajo.reshape(tr.shape), f"trajectory on rank {r}")
        X, ANC, LW, _ = s.eng.traces()
        _eq(X, Xo[:, r * Nl:(r + 1) * Nl], f"state_trace shard {r}")
        _eq(ANC[: pb.T - 1], ANCo[:, r * Nl:(r + 1) * Nl], f"ancestor_trace shard {r}")
        _eq(LW, lwo[r * Nl:(r + 1) * Nl], f"log_weights shard {r}")


def test_config3_geometry_eight_million_particles():
    """BASELINE configs[3]'s exact geometry -- N = 2^23 = 8192 segments = 128 groups = two FULL top-level blocks, which is also the
    engine's capacity (PG_MAX_NSEG, PG_MAX_GRP = 128) -- executed both ways against ONE run of the canonical oracle, bit for bit:
    (a) 8 shards x 1024 segments (the partition of the 8-GPU run, emulated on one device: every rank's window is 1/8 of the CDF,
        most ancestors are remote), traces and trajectory on every rank;
    (b) one unsharded context at the pgas_create limit itself (k_step grid of 8193 workgroups), sweep and step API."""
    from pgas_amd import sharded

    world, Nl = 8, 1024 * 1024
    N = world * Nl
    T = 3
    pb = experiments.smo_pgas(T=T)
    A, S = experiments.initial_params(pb)
    cm = canon_model(pb, N)
    LS, LSinv, cS = cm.chol_parts(S)
    L0 = np.linalg.cholesky(pb.init_state_cov)
    trajo, Xo, ANCo, lwo = cm.sweep(SEED, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, L0)
    # (a) sharded 8 x 2^20
    grp = sharded.make_local_group(world, N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn)
    trajs = sharded.sharded_sweep(grp, SEED, pb.X_true, A, S)
    for r, (s, tr) in enumerate(zip(grp.shards, trajs)):
        _eq(tr, trajo.reshape(tr.shape), f"trajectory on rank {r}")
        X, ANC, LW, _ = s.eng.traces()
        _eq(X, Xo[:, r * Nl:(r + 1) * Nl], f"state_trace shard {r}")
        _eq(ANC[: T - 1], ANCo[:, r * Nl:(r + 1) * Nl], f"ancestor_trace shard {r}")
        _eq(LW, lwo[r * Nl:(r + 1) * Nl], f"log_weights shard {r}")
    for s in grp.shards:
        s.eng.close()
    del grp, trajs
    torch.cuda.empty_cache()
    # (b) unsharded, N = 2^23 on one device
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov,
                                             pb.likelihood_fcn, pb.basis_fcn)
    traj = csmc(SEED, pb.X_true, A, S)
    X, ANC, LW, _ = csmc.engine.traces()
    _eq(X, Xo, "state_trace at N=2^23")
    _eq(ANC[: T - 1], ANCo, "ancestor_trace at N=2^23")
    _eq(LW, lwo, "log_weights at N=2^23")
    _eq(traj, trajo.reshape(traj.shape), "trajectory at N=2^23")
    # step API at the same size: step 2 from the oracle's own step-1 state (teacher-forced)
    lw1, x1, _ = cm.step(1, SEED, Xo[0], None, A, LS, LSinv, cS, pb.X_true[1])
    lw2, x2, a2 = cm.step(2, SEED, x1, lw1, A, LS, LSinv, cS, pb.X_true[2])
    lwg, xg, ag = csmc.step(SEED, 2, torch.as_tensor(lw1), torch.as_tensor(x1), A, S, pb.X_true[2])
    _eq(xg, x2, "step API new_state at N=2^23")
    _eq(ag, a2, "step API a_indices at N=2^23")
    _eq(lwg, lw2, "step API new_log_weights at N=2^23")


def test_capacity_limits_are_clean_errors():
    """One particle beyond the capacity (8193 segments) is refused by pgas_create with a message, and so is a shard layout whose global
    segment count exceeds it -- never a silent wrap of the 128-group top level."""
    from pgas_amd import sharded
    from pgas_amd._lib import PgasError

    pb = experiments.smo_pgas(T=3)
    with pytest.raises(PgasError, match="exceeds"):
        pgas_amd.condSequentialMonteCarlo((1 << 23) + 1, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn)
    with pytest.raises(PgasError, match="exceed"):
        sharded.make_local_group(8, 8 * 1025 * 1024, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn)


def test_filtering_free_functions():
    """src/Filtering.py mirror: systematic_SISR KATs and reconstruct_trajectory on a hand-built ancestry."""
    from oracle import pgas_numpy as o
    from pgas_amd import random as prng
    from pgas_amd.Filtering import STREAM_SISR

    key = 7
    u = float(prng.uniform(key, 1, stream=STREAM_SISR)[0])
    w = np.array([0.1, 0.2, 0.3, 0.4])
    assert pgas_amd.systematic_SISR(key, w).cpu().numpy().tolist() == o.systematic_SISR(u, w).tolist()
    assert pgas_amd.systematic_SISR(key, np.zeros(9)).cpu().numpy().tolist() == list(range(9))          # Filtering.py:25
    assert pgas_amd.systematic_SISR(key, [-1.0, 0.5, 0.5]).cpu().numpy().tolist() == o.systematic_SISR(u, [0.0, 0.5, 0.5]).tolist()
    rng = np.random.default_rng(3)
    for N in (1000, 5000, 70001):
        w = rng.random(N) ** 4
        got = pgas_amd.systematic_SISR(key, w).cpu().numpy()
        ref = o.systematic_SISR(u, w)
        assert np.all(np.diff(got) >= 0)
        bad = np.nonzero(got != ref)[0]
        W = np.cumsum(w / w.sum())
        U = (u + np.arange(N)) / N
        assert all(abs(W[min(got[i], ref[i])] - U[i]) < 1e-9 for i in bad), "indices differ away from a CDF tie"
        assert np.all(np.abs(np.bincount(got, minlength=N) - N * w / w.sum()) <= 1 + 1e-6)
    P = np.arange(12, dtype=float).reshape(4, 3)
    anc = np.array([[2, 0, 1], [1, 1, 0], [0, 2, 2]], dtype=float)   # float64 like the reference's trace (Q2)
    assert pgas_amd.reconstruct_trajectory(P[:, :, None], anc, 1).cpu().numpy().tolist() == [2.0, 3.0, 8.0, 10.0]
    X = rng.standard_normal((6, 500, 2))
    A = rng.integers(0, 500, (5, 500))
    assert np.array_equal(pgas_amd.reconstruct_trajectory(X, A, 123).cpu().numpy(), o.reconstruct_trajectory(X, A, 123))


@pytest.mark.parametrize("N,T", [(1, 5), (2, 4), (63, 3), (1024, 1), (300, 2)])
def test_tiny_sizes(N, T):
    """Edge sizes: a single particle (only the conditioned one), fewer particles than a wave, T = 1 (no step at all), T = 2."""
    pb = experiments.smo_pgas(T=max(T, 2))
    if T == 1:
        pb.observations, pb.inputs, pb.X_true = pb.observations[:1], pb.inputs[:1], pb.X_true[:1]
    A, S = experiments.initial_params(experiments.smo_pgas(T=8))
    cm = canon_model(pb, N)
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov,
                                             pb.likelihood_fcn, pb.basis_fcn)
    LS, LSinv, cS = cm.chol_parts(S)
    traj = csmc(SEED, pb.X_true, A, S)
    if T == 1:
        x0 = cm.init_state(SEED, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov), pb.X_true[0])
        idx = cm.final_index(SEED, np.zeros(N))
        assert csmc.engine.last_final_index() == idx
        _eq(traj, x0[idx].reshape(traj.shape), "T=1 trajectory")
        return
    trajo, Xo, ANCo, lwo = cm.sweep(SEED, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov))
    X, ANC, LW, _ = csmc.engine.traces()
    _eq(X, Xo, "state_trace")
    _eq(ANC[: pb.T - 1], ANCo, "ancestor_trace")
    _eq(traj, trajo.reshape(traj.shape), "trajectory")


@pytest.mark.parametrize("name,N", [("smo", 3000), ("toy", 1500), ("veh27", 2500), ("smo", 70000)])
def test_corrected_mode_bit_exact(name, N):
    """PGAS_OPT_RESAMPLE_BEFORE_PROPAGATE (corrected mode, not the reference's behaviour): step and whole sweep against the
    canonical oracle running the same mode."""
    pb = _problems()[name]()
    A, S = experiments.initial_params(pb)
    cm = canon_model(pb, N)
    cm.set_corrected(True)
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov,
                                             pb.likelihood_fcn, pb.basis_fcn, resample_before_propagate=True)
    LS, LSinv, cS = cm.chol_parts(S)
    L0 = np.linalg.cholesky(pb.init_state_cov)
    x = cm.init_state(SEED, pb.init_state_mean, L0, pb.X_true[0])
    lw = np.zeros(N)
    for t in (1, 2, 3):
        lwc, xc, ac = cm.step(t, SEED, x, lw, A, LS, LSinv, cS, pb.X_true[t])
        lwg, xg, ag = csmc.step(SEED, t, torch.as_tensor(lw), torch.as_tensor(x), A, S, pb.X_true[t])
        _eq(ag, ac, f"corrected ancestors t={t}")
        _eq(xg, xc, f"corrected state t={t}")
        _eq(lwg, lwc, f"corrected log-weights t={t}")
        lw, x = lwc, xc
    traj, X, ANC, lwl = cm.sweep(SEED, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, L0)
    tg = csmc(SEED, pb.X_true, A, S)
    xt, at, lwt, _ = csmc.engine.traces()
    _eq(at, ANC, "corrected sweep ancestors")
    _eq(xt, X, "corrected sweep states")
    _eq(lwt, lwl, "corrected sweep final log-weights")
    _eq(tg, traj.reshape(tuple(tg.shape)), "corrected sweep trajectory")
    assert (ANC[:, :-1] != np.arange(N - 1)).any()
