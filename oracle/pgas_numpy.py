"""NumPy/SciPy fp64 restatement of the reference's particle-filter hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may.

PARITY UNPINNED against the JAX reference: the reference (JAX 0.4.38 + equinox 0.12.2) cannot be
imported offline, ships no tests, no golden vectors and no stored results, so
this restatement is pinned only by the analytic known-answer tests in tests/test_oracle_numpy.py
(SURVEY.md section 8c list) and by SciPy as an independent implementation of the standard
definitions.  Randomness is an explicit input everywhere (JAX's threefry streams cannot be
regenerated here); "identical seeds" is realised between this oracle, the canonical C oracle and
the HIP engine, all of which consume the same Philox-derived numbers.

Each function cites the reference file:line it follows (paths relative to /root/reference).
Statement order follows the reference so that fp64 rounding matches what XLA-CPU would do up to
the usual libm/reduction-order differences.
"""
from __future__ import annotations

import heapq
import math

import numpy as np
import scipy.linalg as sla
import scipy.special as ssp

LOG_2PI = math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------------------------
# src/Filtering.py
# ----------------------------------------------------------------------------------------------
def systematic_SISR(u, w):
    """src/Filtering.py:6-37.  `u` replaces jax.random.uniform(key) (:19)."""
    w = np.asarray(w, dtype=np.float64)
    N = len(w)
    w = np.clip(w, 0.0, np.inf)  # :23
    w_sum = np.sum(w)  # :24
    w = w / w_sum if w_sum > 0 else np.ones_like(w) / N  # :25
    U = (u + np.arange(N)) / N  # :28
    W = np.clip(np.cumsum(w), 0.0, 1.0)  # :29-32
    idx = np.searchsorted(W, U, side="left")  # :34 (jnp default side='left')
    return np.clip(idx, 0, N - 1).astype(np.int32)  # :35


def reconstruct_trajectory(Particles, ancestry, idx):
    """src/Filtering.py:40-55 (host back-trace)."""
    P = np.atleast_3d(Particles)
    T, n = P.shape[0], P.shape[-1]
    traj = np.zeros((T, n))
    b = int(idx)
    traj[T - 1] = P[T - 1, b]
    for i in range(T - 2, -1, -1):
        b = int(ancestry[i, b])
        traj[i] = P[i, b]
    return np.squeeze(traj)


# ----------------------------------------------------------------------------------------------
# src/BasisFunctions.py
# ----------------------------------------------------------------------------------------------
def hilbert_index_table(num_fcn, domain_boundary, idx_start=1, idx_step=1):
    """Index tuples of the num_fcn Laplacian eigenfunctions with the smallest eigenvalue.

    src/BasisFunctions.py:12-59: best-first search on the index lattice with a min-heap keyed by
    (cost, lattice position); the cost of a neighbour is the parent's cost plus the float
    increment weights[d]*(j^2[new]-j^2[old]) (:52-55), and ties are broken by tuple order.
    Returns (S (M,D) float array of frequencies j, domain_size (D,), domain_center (D,)).
    """
    box = np.atleast_2d(np.asarray(domain_boundary, dtype=np.float64))
    D = box.shape[0]
    center = (box[:, 0] + box[:, 1]) / 2  # :16
    if idx_start < 1:  # :19-20
        idx_start = 1
    size = box[:, 1] - box[:, 0]  # :23
    j = np.arange(idx_start, num_fcn * idx_step + 1 + idx_start, idx_step)  # :24-25
    wgt = (np.pi / size) ** 2  # :29
    jsq = j**2  # :30
    origin = (0,) * D
    frontier = [(float(np.sum(wgt * jsq[0])), origin)]  # :33-35
    seen = {origin}
    chosen = []
    while len(chosen) < num_fcn and frontier:  # :39
        cost, pos = heapq.heappop(frontier)
        chosen.append(j[np.array(pos, dtype=int)])
        for d in range(D):  # :45
            if pos[d] + 1 >= len(j):
                continue
            nxt = pos[:d] + (pos[d] + 1,) + pos[d + 1 :]
            if nxt in seen:
                continue
            inc = float(wgt[d] * (jsq[nxt[d]] - jsq[pos[d]]))  # :53-55
            heapq.heappush(frontier, (cost + inc, nxt))
            seen.add(nxt)
    S = np.array(chosen, dtype=float)  # :59
    return S, size, center


def spectral_density_Gaussian(freq, magnitude, lengthscale):
    """src/BasisFunctions.py:83-105 for one frequency vector."""
    freq = np.asarray(freq, dtype=np.float64)
    D = len(freq)
    ls = np.broadcast_to(lengthscale, freq.shape)
    return magnitude * (2 * np.pi) ** (D / 2) * np.prod(ls) * np.exp(-0.5 * np.sum(ls**2 * freq**2))


def generate_Hilbert_BasisFunction(num_fcn, domain_boundary, lengthscale, scale, idx_start=1, idx_step=1):
    """src/BasisFunctions.py:8-74.  Returns (phi, spectral_density, S) -- S is extra."""
    S, size, center = hilbert_index_table(num_fcn, domain_boundary, idx_start, idx_step)
    eig = (np.pi * S / size) ** 2  # :60
    L = size / 2

    def phi(x):
        # :63-66 and _eigen_fnc :77-80; scalar x broadcasts for D = 1
        xc = np.asarray(x, dtype=np.float64) - center
        return np.prod(np.sqrt(1 / L) * np.sin(np.sqrt(eig) * (xc + L)), axis=1)

    sd = np.array([spectral_density_Gaussian(f, scale, lengthscale) for f in np.sqrt(eig)])  # :69-72
    return phi, sd, S


# ----------------------------------------------------------------------------------------------
# src/BayesianInferrence.py
# ----------------------------------------------------------------------------------------------
def _solve_spd(A, B):
    """BI:11-13."""
    return sla.cho_solve((np.linalg.cholesky(A), True), B)


def prior_mniw_2naturalPara(mean, col_cov, row_scale, df):
    """BI:18-32."""
    mean = np.atleast_2d(mean)
    row_scale = np.atleast_2d(row_scale)
    tmp = _solve_spd(col_cov, np.hstack([mean.T, np.eye(col_cov.shape[0])]))
    eta_0 = tmp[:, : mean.shape[0]]
    eta_1 = tmp[:, mean.shape[0] :]
    eta_2 = mean @ eta_0 + row_scale
    return eta_0, eta_1, eta_2, df


def prior_mniw_2naturalPara_inv(eta_0, eta_1, eta_2, eta_3):
    """BI:35-45."""
    tmp = _solve_spd(eta_1, np.hstack([eta_0, np.eye(eta_1.shape[0])]))
    mean = tmp[:, : eta_0.shape[1]].T
    col_cov = tmp[:, eta_0.shape[1] :]
    row_scale = eta_2 - mean @ eta_0
    return np.atleast_2d(mean), col_cov, np.atleast_2d(row_scale), eta_3


def prior_mniw_mean(eta_0, eta_1):
    """BI:48-50."""
    return _solve_spd(0.5 * (eta_1 + eta_1.T), eta_0).T


def prior_mniw_calcStatistics(y, basis):
    """BI:53-61."""
    return np.outer(basis, y), np.outer(basis, basis), np.outer(y, y), 1


def prior_mniw_Predictive(mean, col_cov, row_scale, df, basis):
    """BI:64-89."""
    basis = np.atleast_2d(basis)
    col_cov = np.atleast_2d(col_cov)
    row_scale = np.atleast_2d(row_scale)
    df = df + 1 - row_scale.shape[0]
    m = np.squeeze(basis @ mean.T)
    col_scale = basis @ col_cov @ basis.T + np.eye(basis.shape[0])
    return m, col_scale, row_scale / df, df


def prior_mniw_log_base_measure(T_0, T_1, T_2, T_3):
    """BI:111-124."""
    n, m = T_2.shape[0], T_1.shape[0]
    Psi = T_2 - T_0.T @ _solve_spd(T_1, T_0)
    nu = T_3
    return (
        -0.5 * n * m * np.log(2 * np.pi)
        + 0.5 * n * np.log(np.linalg.det(T_1))
        - 0.5 * nu * n * np.log(2)
        - ssp.multigammaln(nu / 2, n)
        + np.log(np.linalg.det(Psi)) * nu / 2
    )


# ----------------------------------------------------------------------------------------------
# helpers standing in for jax library calls
# ----------------------------------------------------------------------------------------------
def softmax(x):
    """jax.nn.softmax: exp(x - max) / sum."""
    e = np.exp(x - np.max(x))
    return e / np.sum(e)


def mvn_logpdf(x, mean, cov):
    """jax.scipy.stats.multivariate_normal.logpdf (Cholesky form); mean may be (N,n)."""
    cov = np.atleast_2d(cov)
    n = cov.shape[0]
    L = np.linalg.cholesky(cov)
    d = np.atleast_1d(x) - np.atleast_2d(mean)
    y = sla.solve_triangular(L, d.T, lower=True)
    return -0.5 * n * LOG_2PI - np.sum(np.log(np.diag(L))) - 0.5 * np.sum(y**2, axis=0)


# ----------------------------------------------------------------------------------------------
# src/PGAS.py
# ----------------------------------------------------------------------------------------------
class condSequentialMonteCarlo:
    """src/PGAS.py:14-228 with explicit random inputs.

    likelihood_fcn(obs, state(N,nx), input) -> (N,) and basis_fcn(state(N,nx), input) -> (N,M) are
    vectorised over particles (the reference vmaps scalar callables, :52,:93,:138).
    """

    def __init__(self, N_samples, observations, inputs, init_state_mean, init_state_cov, likelihood_fcn, basis_fcn):
        self.N = N_samples
        self.y = np.asarray(observations, dtype=np.float64)
        self.u = np.asarray(inputs, dtype=np.float64)
        self.m0 = np.asarray(init_state_mean, dtype=np.float64)
        self.P0 = np.asarray(init_state_cov, dtype=np.float64)
        self.lik = likelihood_fcn
        self.basis = basis_fcn

    def step(self, u_resample, u_ancestor, z, time, log_weights, state, coeff_mat, error_cov, ref_state):
        """:79-153.  z (N,nx) standard normals replace the per-particle mvn draw (:72-75)."""
        Phi = self.basis(state, self.u[time])  # :52-54  (Q3: u_t with x_{t-1})
        aux = Phi @ coeff_mat.T  # :55
        ll_aux = self.lik(self.y[time], aux, self.u[time])  # :93-100
        lw_aux = ll_aux + log_weights  # :101
        a = systematic_SISR(u_resample, softmax(lw_aux))  # :102-106
        h = mvn_logpdf(ref_state, aux, error_cov)  # :109-116
        w_anc = softmax(lw_aux + h)  # :117-118
        ref_idx = int(np.searchsorted(np.cumsum(w_anc), u_ancestor))  # :122-124
        ref_idx = min(ref_idx, self.N - 1)  # Q4: JAX clamps the out-of-range gather at :146
        a = a.copy()
        a[-1] = ref_idx  # :127
        Ls = np.linalg.cholesky(np.atleast_2d(error_cov))
        if getattr(self, "resample_before_propagate", False):
            new_state = aux[a] + z @ Ls.T  # CORRECTED mode (not the reference): propagate state[a], cf. src/Algorithm1.py:286-292
        else:
            new_state = aux + z @ Ls.T  # :130-133 (Q1: from `state`, not state[a]; Q6: same Phi)
        new_state[-1] = ref_state  # :134
        new_lw = self.lik(self.y[time], new_state, self.u[time]) - ll_aux[a]  # :137-147
        return new_lw, new_state, a

    def init_state(self, z0, ref0):
        """:155-174 and :194."""
        L0 = np.linalg.cholesky(self.P0)
        x0 = self.m0 + z0 @ L0.T
        x0[-1] = ref0
        return x0

    def __call__(self, rand, ref_state, coeff_mat, error_cov, return_traces=False):
        """:176-228.  `rand` supplies z0 (N,nx), z (T,N,nx), u_resample (T,), u_ancestor (T,), u_final."""
        T = self.y.shape[0]
        nx = self.m0.shape[0]
        ref = np.asarray(ref_state, dtype=np.float64).reshape(T, nx)
        X = np.zeros((T, self.N, nx))
        LW = np.zeros((T, self.N))
        ANC = np.zeros((T, self.N))  # Q2: float64, last row unused
        X[0] = self.init_state(rand["z0"], ref[0])
        for t in range(1, T):  # :199
            LW[t], X[t], ANC[t - 1] = self.step(
                rand["u_resample"][t], rand["u_ancestor"][t], rand["z"][t], t, LW[t - 1], X[t - 1], coeff_mat, error_cov, ref[t]
            )
        w = softmax(LW[-1])  # :224
        idx = min(int(np.searchsorted(np.cumsum(w), rand["u_final"])), self.N - 1)  # :225
        traj = reconstruct_trajectory(X, ANC, idx)  # :226
        if return_traces:
            return traj, X, LW, ANC, idx
        return traj


def suff_stats(traj, Phi):
    """src/PGAS.py:294-303 without the prior: sums of BI:53-61 over t."""
    traj = np.asarray(traj, dtype=np.float64)
    traj = traj.reshape(traj.shape[0], -1)
    Xp = traj[1:]
    T0 = Phi.T @ Xp
    T1 = Phi.T @ Phi
    T2 = Xp.T @ Xp
    return T0, T1, T2, float(Xp.shape[0])


def sample_params(GP_prior, T0, T1, T2, T3, chi2_draws, normals_T, normals_A):
    """src/PGAS.py:298-343.  chi2_draws (p,) ~ chi^2(df - i), normals_T (p,p), normals_A (n_x,M)."""
    stats = (GP_prior[0] + T0, GP_prior[1] + T1, GP_prior[2] + T2, GP_prior[3] + T3)
    mean, col_cov, row_scale, df = prior_mniw_2naturalPara_inv(*stats)  # :306
    p = row_scale.shape[0]
    chol_row = np.linalg.cholesky(row_scale)  # :317
    L = sla.solve_triangular(chol_row, np.eye(p), lower=True)  # :319
    Tm = np.tril(normals_T, k=-1) + np.diag(np.sqrt(chi2_draws))  # :327-329
    C = L @ Tm  # :332
    S_chol = sla.solve_triangular(C.T, np.eye(p), lower=False)  # :334
    S = S_chol @ S_chol.T  # :335
    V_chol = np.linalg.cholesky(col_cov)  # :339
    A = mean + S_chol @ normals_A @ V_chol  # :341 (Q5)
    return A, S, (mean, col_cov, row_scale, df)


def pgas_chain(sweep_fcn, basis_fcn, likelihood_fcn, GP_prior, observations, inputs, init_ref_state, K, step_seeds, draws, params=None):
    """PGAS.__call__, src/PGAS.py:345-397, with explicit randomness.

    sweep_fcn(seed, ref (T,nx), A, S) -> traj (T,nx) stands for condSequentialMonteCarlo.__call__ (:366-371; this module's class
    with the canonical random numbers, or the canonical C oracle).  step_seeds[k] is the key handed to sweep k (:365),
    draws[k] = dict(chi2, normals_T, normals_A) the numbers the k-th sample_params consumes (k = 0: :358, k >= 1: :378).
    basis_fcn(states (n,nx), input) -> (n,M) and likelihood_fcn(obs, states (n,nx), input) -> (n,) are vectorised over rows.

    `params`, if given, is a list of K pairs (A_k, S_k): sweep k+1 then runs with params[k] instead of this restatement's own
    draw -- teacher forcing, so that two implementations whose sample_params agree to rounding can still be compared sweep
    by sweep (a sweep is only reproducible bit for bit from identical (A, S)).

    Returns state_trace (T,K,nx) (:380), log_likelihood (T,K) (:383-392) and the restatement's own [(A_k, S_k)].
    """
    y = np.asarray(observations, dtype=np.float64)
    T = y.shape[0]
    y = y.reshape(T, -1)
    u = np.asarray(inputs, dtype=np.float64).reshape(T, -1)
    ref0 = np.asarray(init_ref_state, dtype=np.float64).reshape(T, -1)
    nx = ref0.shape[1]
    trace = np.zeros((K, T, nx))                                   # :266-273
    trace[0] = ref0                                                # :354

    def draw_params(k):
        traj = trace[k]
        Phi = np.vstack([basis_fcn(traj[t:t + 1], u[t]) for t in range(T - 1)])   # :294-296 (Q3: traj[:-1] with inputs[:-1])
        T0, T1, T2, T3 = suff_stats(traj, Phi)                                     # :297-303
        d = draws[k]
        A, S, _ = sample_params(GP_prior, T0, T1, T2, T3, d["chi2"], d["normals_T"], d["normals_A"])   # :306-341
        return A, S

    own = [draw_params(0)]                                         # :358
    for k in range(1, K):                                          # :361
        A, S = params[k - 1] if params is not None else own[k - 1]
        trace[k] = np.asarray(sweep_fcn(step_seeds[k], trace[k - 1], A, S), dtype=np.float64).reshape(T, nx)   # :366-374
        own.append(draw_params(k))                                 # :378
    state_trace = np.swapaxes(trace, 0, 1)                         # :380 -> (T,K,nx)
    ll = np.empty((T, K))
    for t in range(T):                                             # :383-392
        ll[t] = likelihood_fcn(y[t], state_trace[t], u[t])
    return state_trace, ll, own
