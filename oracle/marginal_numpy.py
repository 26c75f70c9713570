"""CPU restatement (NumPy, fp64) of the reference's MARGINALISED particle-filter family -- TEST INFRASTRUCTURE ONLY.

    src/StateSpaceModel.py:8-87      StateSpaceModel
    src/Algorithm1.py:27-492         Algorithm1  (online, per-particle MNIW sufficient statistics)
    src/Algorithm3.py:15-303         Algorithm3  (conditional version with ancestor sampling)
    src/Algorithm2.py:12-187         Algorithm2  (Particle Gibbs over Algorithm3)
    src/BayesianInferrence.py:48-124 prior_mniw_mean / Predictive / drawPred / log_base_measure (via oracle/pgas_numpy.py)

PARITY UNPINNED: as for oracle/pgas_numpy.py, the JAX reference cannot run here and ships no outputs; this file follows the
reference line by line (citations at each step) and is pinned only by analytic checks (tests/test_marginal_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Differences of form, not of arithmetic: the reference maps per-particle functions with jax.vmap, here every callable is
vectorised over the leading particle axis; random numbers are an explicit provider `rand` (see CanonRand in tests/common.py)
instead of jax.random keys; the Student-t draw of BI:104 is taken from the provider with the df the reference computes.
"""
from __future__ import annotations

import numpy as np
from scipy.special import multigammaln

from . import pgas_numpy as o

STREAM_INIT_STATE, STREAM_STATE, STREAM_RESAMPLE, STREAM_ANCESTOR, STREAM_FINAL, STREAM_INIT_INTVAR, STREAM_INTVAR = 16, 17, 18, 19, 20, 24, 32


class StateSpaceModel:
    """src/StateSpaceModel.py:8-87; transition_model / output_model take (state (N,nx), input (nu,), *int_var (N,n_i))."""

    def __init__(self, process_noise, output_noise, transition_model, output_model):
        self.process_noise = np.atleast_2d(np.asarray(process_noise, dtype=np.float64))
        self.output_noise = np.atleast_2d(np.asarray(output_noise, dtype=np.float64))
        self.transition_model = transition_model
        self.output_model = output_model
        self.is_deterministic = bool(np.all(self.process_noise == 0))  # :30

    def transition_mdl(self, state, input, *int_var):  # :32-42
        return self.transition_model(state, input, *int_var)

    def output_mdl(self, state, input, *int_var):  # :44-54
        return self.output_model(state, input, *int_var)

    def draw_state(self, z, state, input, *int_var):  # :56-74, z (N,nx) standard normals
        new_state = self.transition_mdl(state, input, *int_var)
        if self.is_deterministic:
            return new_state
        return new_state + z @ np.linalg.cholesky(self.process_noise).T

    def log_likelihood(self, observation, state, input, *int_var):  # :76-87
        out = np.asarray(self.output_mdl(state, input, *int_var), dtype=np.float64)
        out = out.reshape(out.shape[0], -1)
        return o.mvn_logpdf(np.atleast_1d(observation), out, self.output_noise)


def _calc_stats(int_var, basis):
    """vmap(BI.prior_mniw_calcStatistics) (BI:53-61): int_var (N,n), basis (N,M) -> T0 (N,M,n), T1 (N,M,M), T2 (N,n,n), T3 (N,)."""
    int_var = np.asarray(int_var, dtype=np.float64).reshape(basis.shape[0], -1)
    return (np.einsum("nm,nk->nmk", basis, int_var), np.einsum("nm,nl->nml", basis, basis),
            np.einsum("nk,nl->nkl", int_var, int_var), np.ones(basis.shape[0]))  # Q12: T3 = 1 per particle


def _solve_spd_b(A, B):
    L = np.linalg.cholesky(A)
    y = np.linalg.solve(L, B)
    return np.linalg.solve(np.swapaxes(L, -1, -2), y)


def _mniw_mean_b(eta0, eta1):  # vmap(BI.prior_mniw_mean) (BI:48-50) -> (N,n,M)
    sym = 0.5 * (eta1 + np.swapaxes(eta1, -1, -2))
    return np.swapaxes(_solve_spd_b(sym, eta0), -1, -2)


def _natural_inv_b(eta0, eta1, eta2, eta3):  # vmap(BI.prior_mniw_2naturalPara_inv) (BI:35-45)
    N, M, n = eta0.shape
    sol = _solve_spd_b(eta1, np.concatenate([eta0, np.broadcast_to(np.eye(M), (N, M, M))], axis=2))
    mean = np.swapaxes(sol[:, :, :n], -1, -2)          # (N,n,M)
    col_cov = sol[:, :, n:]                             # (N,M,M)
    row_scale = eta2 - mean @ eta0                      # (N,n,n)
    return mean, col_cov, row_scale, eta3


def _log_base_measure_b(T0, T1, T2, T3):  # vmap(BI.prior_mniw_log_base_measure) (BI:111-124)
    n, m = T2.shape[-1], T1.shape[-1]
    Psi = T2 - np.swapaxes(T0, -1, -2) @ _solve_spd_b(T1, T0)
    nu = T3
    return (-0.5 * n * m * np.log(2 * np.pi) + 0.5 * n * np.log(np.linalg.det(T1)) - 0.5 * nu * n * np.log(2)
            - multigammaln(nu / 2, n) + np.log(np.linalg.det(Psi)) * nu / 2)


class Algorithm1:
    """src/Algorithm1.py:27-492.  basis_fcn[i](state (N,nx), input) -> (N,M_i); GP_prior[i] = (eta0 (M,n), eta1 (M,M), eta2 (n,n), eta3)."""

    def __init__(self, N_samples, observations, inputs, SSM, forgetting_factor, init_state_mean, init_state_cov, init_int_var_mean,
                 init_int_var_cov, GP_prior, basis_fcn):
        self.N_samples = int(N_samples)
        self.observations = np.asarray(observations, dtype=np.float64)
        self.inputs = np.asarray(inputs, dtype=np.float64)
        self.SSM = SSM
        self.forgetting_factor = float(forgetting_factor)
        self.init_state_mean = np.asarray(init_state_mean, dtype=np.float64).reshape(-1)
        self.init_state_cov = np.atleast_2d(np.asarray(init_state_cov, dtype=np.float64))
        self.init_int_var_mean = [np.asarray(m, dtype=np.float64).reshape(-1) for m in init_int_var_mean]
        self.init_int_var_cov = [np.atleast_2d(np.asarray(c, dtype=np.float64)) for c in init_int_var_cov]
        self.basis_fcn = list(basis_fcn)
        self.GP_prior = [[np.asarray(g[0], dtype=np.float64), np.asarray(g[1], dtype=np.float64),
                          np.atleast_2d(np.asarray(g[2], dtype=np.float64)), float(g[3])] for g in GP_prior]
        self.N_int = len(self.basis_fcn)

    # ---- :100-177
    def _init_algorithm(self, rand):
        N, T = self.N_samples, self.observations.shape[0]
        nx = self.init_state_mean.shape[0]
        state_trace = np.zeros((T, N, nx))
        int_var_trace = [np.zeros((T, N, m.shape[0])) for m in self.init_int_var_mean]
        suff_stats_trace = [[np.zeros((T, *g[0].shape)), np.zeros((T, *g[1].shape)), np.zeros((T, *g[2].shape)), np.zeros(T)] for g in self.GP_prior]
        log_weights_trace = np.zeros((T, N))
        ancestor_trace = np.zeros((T - 1, N), dtype=np.int32)
        state_trace[0] = self.init_state_mean + rand.normal(STREAM_INIT_STATE, 0, nx) @ np.linalg.cholesky(self.init_state_cov).T  # :139-145
        suff_stats = []
        for i in range(self.N_int):
            n = self.init_int_var_mean[i].shape[0]
            int_var_trace[i][0] = self.init_int_var_mean[i] + rand.normal(STREAM_INIT_INTVAR + i, 0, n) @ np.linalg.cholesky(self.init_int_var_cov[i]).T  # :146-153
            basis = self.basis_fcn[i](state_trace[0], self.inputs[0])                                 # :158-160
            Ts = _calc_stats(int_var_trace[i][0], basis)                                             # :161-163
            suff_stats.append(Ts)
            w = o.softmax(log_weights_trace[0])                                                       # :166
            for j in range(4):
                suff_stats_trace[i][j][0] = np.einsum("j...,j->...", Ts[j], w)                       # :167-170
        return state_trace, int_var_trace, suff_stats_trace, log_weights_trace, ancestor_trace, tuple(suff_stats)

    # ---- :179-232
    def _generate_auxiliary_states(self, state, time, int_var, suff_stats):
        aux_state = self.SSM.transition_mdl(state, self.inputs[time - 1], *int_var)                  # :206-208
        aux_int_var = []
        for i in range(self.N_int):
            mean = _mniw_mean_b(suff_stats[i][0] + self.GP_prior[i][0], suff_stats[i][1] + self.GP_prior[i][1])   # :211-217
            basis = self.basis_fcn[i](aux_state, self.inputs[time])                                   # :220-225
            aux_int_var.append(np.einsum("ikj,ij->ik", mean, basis))                                  # :228-231
        return aux_state, tuple(aux_int_var)

    # ---- :234-273
    def _draw_int_vars(self, rand, time, state, suff_stats):
        int_var, basis_all = [], []
        for i in range(self.N_int):
            basis = self.basis_fcn[i](state, self.inputs[time])                                       # :243-248
            mean, col_cov, row_scale, df = _natural_inv_b(*[suff_stats[i][j] + self.GP_prior[i][j] for j in range(4)])  # :251-256
            n = row_scale.shape[-1]
            dfp = df + 1 - n                                                                          # BI:78
            pmean = np.einsum("nkm,nm->nk", mean, basis)                                              # BI:81
            col_scale = np.einsum("nm,nml,nl->n", basis, col_cov, basis) + 1.0                        # BI:84 (n_b = 1)
            prow = row_scale / np.reshape(dfp, (-1, 1, 1))                                            # BI:87
            t = rand.student_t(STREAM_INTVAR + i, time, np.broadcast_to(dfp, (state.shape[0],)).copy(), n)   # BI:104, (N,n)
            Lr = np.linalg.cholesky(prow)
            draw = pmean + np.einsum("nij,nj->ni", Lr, t) * np.sqrt(col_scale)[:, None]              # BI:106-108
            int_var.append(draw)
            basis_all.append(basis)
        return tuple(int_var), tuple(basis_all)

    # ---- :275-295
    def _draw_states(self, rand, time, state, int_var, a):
        z = rand.normal(STREAM_STATE, time, state.shape[1])
        return self.SSM.draw_state(z, state[a], self.inputs[time - 1], *[v[a] for v in int_var])

    # ---- :297-397
    def step(self, rand, time, log_weights, state, int_var, suff_stats):
        lam = self.forgetting_factor
        suff_stats = tuple(tuple(s * lam for s in suff_stats[i]) for i in range(self.N_int))         # :317-320
        aux_state, aux_int_var = self._generate_auxiliary_states(state, time, int_var, suff_stats)    # :323-325
        ll_aux = self.SSM.log_likelihood(self.observations[time], aux_state, self.inputs[time], *aux_int_var)   # :328-341
        a = o.systematic_SISR(rand.uniform(STREAM_RESAMPLE, time), o.softmax(ll_aux + log_weights))  # :342-347
        new_state = self._draw_states(rand, time, state, int_var, a)                                  # :350-353
        args = tuple(tuple(suff_stats[i][j][a] for j in range(4)) for i in range(self.N_int))         # :358-361
        new_int_var, new_basis = self._draw_int_vars(rand, time, new_state, args)                     # :362-367
        new_stats = []
        for i in range(self.N_int):
            Ts = _calc_stats(new_int_var[i], new_basis[i])                                            # :370-373
            new_stats.append(tuple(args[i][j] + Ts[j] for j in range(4)))                             # :374-377
        new_lw = self.SSM.log_likelihood(self.observations[time], new_state, self.inputs[time], *new_int_var) - ll_aux[a]   # :380-390
        return new_lw, new_state, new_int_var, tuple(new_stats), a

    # ---- :399-492
    def __call__(self, rand):
        state_trace, int_var_trace, sst, lw_trace, anc_trace, suff_stats = self._init_algorithm(rand)
        T = self.observations.shape[0]
        for time in range(1, T):
            lw, x, iv, suff_stats, a = self.step(rand, time, lw_trace[time - 1], state_trace[time - 1],
                                                 [int_var_trace[i][time - 1] for i in range(self.N_int)], suff_stats)
            state_trace[time], lw_trace[time], anc_trace[time - 1] = x, lw, a
            w = o.softmax(lw)
            for i in range(self.N_int):
                int_var_trace[i][time] = iv[i]
                for j in range(4):
                    sst[i][j][time] = np.einsum("n...,n->...", suff_stats[i][j], w)                   # :445-457
        weights_trace = np.stack([o.softmax(r) for r in lw_trace])                                    # :460
        obs_trace = np.stack([np.asarray(self.SSM.output_mdl(state_trace[t], self.inputs[t], *[v[t] for v in int_var_trace])).reshape(self.N_samples, -1)
                              for t in range(T)])                                                     # :463-468
        loglik = np.stack([self.SSM.log_likelihood(self.observations[t], state_trace[t], self.inputs[t], *[v[t] for v in int_var_trace])
                           for t in range(T)])                                                        # :471-481
        return state_trace, int_var_trace, sst, weights_trace, anc_trace, suff_stats, obs_trace, loglik


class Algorithm3(Algorithm1):
    """src/Algorithm3.py:15-303 (forgetting factor fixed to 1.0 and never applied, quirk Q11)."""

    def __init__(self, N_samples, observations, inputs, SSM, init_state_mean, init_state_cov, init_int_var_mean, init_int_var_cov, GP_prior, basis_fcn):
        super().__init__(N_samples, observations, inputs, SSM, 1.0, init_state_mean, init_state_cov, init_int_var_mean, init_int_var_cov, GP_prior, basis_fcn)

    # ---- :43-197
    def step(self, rand, time, log_weights, state, int_var, suff_stats, ref_state, ref_int_var, ref_suff_stats):
        N = self.N_samples
        aux_state, aux_int_var = self._generate_auxiliary_states(state, time, int_var, suff_stats)    # :66-68
        ll_aux = self.SSM.log_likelihood(self.observations[time], aux_state, self.inputs[time], *aux_int_var)   # :71-84
        lw_aux = ll_aux + log_weights
        a = o.systematic_SISR(rand.uniform(STREAM_RESAMPLE, time), o.softmax(lw_aux))                 # :85-90
        if self.SSM.is_deterministic:
            # Q15: with process_noise == 0 (Toy_Example.py:66) the density of :109-116 has a singular covariance; in JAX the Cholesky
            # yields NaN, every ancestor weight is NaN and the index that comes out of searchsorted is implementation-defined.
            # Restated as "the reference particle keeps its own ancestor".
            ref_idx = N - 1
        else:
            g_T, g_t = np.zeros(N), np.zeros(N)
            for i in range(self.N_int):                                                               # :95-108
                P, R, S = self.GP_prior[i], ref_suff_stats[i], suff_stats[i]
                g_T += _log_base_measure_b(P[0] + R[0] + S[0], P[1] + R[1] + S[1], P[2] + R[2] + S[2], P[3] + R[3] + S[3])
                g_t += _log_base_measure_b(P[0] + S[0], P[1] + S[1], P[2] + S[2], P[3] + S[3])
            h_x = o.mvn_logpdf(ref_state, aux_state, self.SSM.process_noise)                          # :109-116
            w_anc = o.softmax(lw_aux + g_t - g_T + h_x)                                               # :117-118
            ref_idx = min(int(np.searchsorted(np.cumsum(w_anc), rand.uniform(STREAM_ANCESTOR, time))), N - 1)   # :121-124 (clip: Q4)
        a = a.copy()
        a[-1] = ref_idx                                                                               # :127
        new_state = self._draw_states(rand, time, state, int_var, a)                                  # :130-133
        new_state[-1] = ref_state                                                                     # :134
        args = tuple(tuple(suff_stats[i][j][a] for j in range(4)) for i in range(self.N_int))         # :139-142
        new_int_var, new_basis = self._draw_int_vars(rand, time, new_state, args)                     # :143-148
        new_int_var = list(new_int_var)
        for i in range(self.N_int):
            new_int_var[i][-1] = np.reshape(ref_int_var[i], -1)                                       # :149-152
        new_stats = []
        for i in range(self.N_int):
            Ts = _calc_stats(new_int_var[i], new_basis[i])                                            # :155-158
            new_stats.append(tuple(args[i][j] + Ts[j] for j in range(4)))                             # :159-162
        new_ref = []
        for i in range(self.N_int):                                                                   # :165-176
            rb = self.basis_fcn[i](np.reshape(ref_state, (1, -1)), self.inputs[time])
            rT = _calc_stats(np.reshape(ref_int_var[i], (1, -1)), rb)
            new_ref.append(tuple(ref_suff_stats[i][j] - rT[j][0] for j in range(4)))
        new_lw = self.SSM.log_likelihood(self.observations[time], new_state, self.inputs[time], *new_int_var) - ll_aux[a]   # :179-189
        return new_lw, new_state, tuple(new_int_var), tuple(new_stats), a, tuple(new_ref)

    # ---- :199-303
    def __call__(self, rand, ref_state, ref_int_var, ref_suff_stats):
        state_trace, int_var_trace, _, lw_trace, anc_trace, suff_stats = self._init_algorithm(rand)
        T = self.observations.shape[0]
        ref_state = np.asarray(ref_state, dtype=np.float64).reshape(T, -1)
        ref_int_var = [np.asarray(v, dtype=np.float64).reshape(T, -1) for v in ref_int_var]
        state_trace[0, -1] = ref_state[0]                                                             # :221
        suff_stats = [list(s) for s in suff_stats]
        ref_suff_stats = [list(r) for r in ref_suff_stats]
        for i in range(self.N_int):
            int_var_trace[i][0, -1] = ref_int_var[i][0]                                               # :224
            ib = self.basis_fcn[i](ref_state[:1], self.inputs[0])                                     # :225
            iT = _calc_stats(ref_int_var[i][:1], ib)                                                  # :226
            for j in range(4):
                suff_stats[i][j] = np.array(suff_stats[i][j])
                suff_stats[i][j][-1] = iT[j][0]                                                       # :228-231
                ref_suff_stats[i][j] = ref_suff_stats[i][j] - iT[j][0]                                # :235-246
        suff_stats = tuple(tuple(s) for s in suff_stats)
        ref_suff_stats = tuple(tuple(r) for r in ref_suff_stats)
        for time in range(1, T):                                                                      # :251-290
            lw, x, iv, suff_stats, a, ref_suff_stats = self.step(
                rand, time, lw_trace[time - 1], state_trace[time - 1], [int_var_trace[i][time - 1] for i in range(self.N_int)], suff_stats,
                ref_state[time], [ref_int_var[i][time] for i in range(self.N_int)], ref_suff_stats)
            state_trace[time], lw_trace[time], anc_trace[time - 1] = x, lw, a
            for i in range(self.N_int):
                int_var_trace[i][time] = iv[i]
        w = o.softmax(lw_trace[-1])                                                                   # :293
        idx = min(int(np.searchsorted(np.cumsum(w), rand.uniform(STREAM_FINAL, 0))), self.N_samples - 1)   # :294
        state_traj = o.reconstruct_trajectory(state_trace, anc_trace, idx)                            # :295
        int_var_traj = tuple(o.reconstruct_trajectory(int_var_trace[i], anc_trace, idx) for i in range(self.N_int))   # :296-299
        return state_traj, int_var_traj, dict(state_trace=state_trace, ancestor_trace=anc_trace, log_weights=lw_trace[-1], idx=idx)


def trajectory_stats(alg, state_traj, int_var_traj):
    """Reference sufficient statistics of a trajectory, summed over time (src/Algorithm2.py:81-93,146-160)."""
    T = alg.observations.shape[0]
    out = []
    st = np.asarray(state_traj, dtype=np.float64).reshape(T, -1)
    for i in range(alg.N_int):
        basis = np.stack([alg.basis_fcn[i](st[t:t + 1], alg.inputs[t])[0] for t in range(T)])
        Ts = _calc_stats(np.asarray(int_var_traj[i], dtype=np.float64).reshape(T, -1), basis)
        out.append([np.sum(Ts[j], axis=0) for j in range(4)])
    return out


class Algorithm2:
    """src/Algorithm2.py:12-187: Particle Gibbs over Algorithm3.  `rands` yields one random-number provider per Gibbs iteration
    (the reference splits its key once per iteration, :121)."""

    def __init__(self, N_samples, N_iterations, observations, inputs, SSM, init_state_mean, init_state_cov, init_int_var_mean, init_int_var_cov,
                 GP_prior, basis_fcn):
        self.N_iterations = int(N_iterations)
        self.N_steps = np.asarray(observations).shape[0]
        self.cSMC = Algorithm3(N_samples, observations, inputs, SSM, init_state_mean, init_state_cov, init_int_var_mean, init_int_var_cov,
                               GP_prior, basis_fcn)                                                   # :28-39

    def __call__(self, rands, init_ref_state, init_ref_int_var):
        c, K, T = self.cSMC, self.N_iterations, self.N_steps
        nx = c.init_state_mean.shape[0]
        state_trace = np.zeros((K, T, nx))                                                            # :46-54
        state_trace[0] = np.asarray(init_ref_state, dtype=np.float64).reshape(T, nx)
        int_var_trace = [np.zeros((K, T, m.shape[0])) for m in c.init_int_var_mean]                   # :56-67
        for i in range(c.N_int):
            int_var_trace[i][0] = np.asarray(init_ref_int_var[i], dtype=np.float64).reshape(T, -1)
        sst = [[np.zeros((K, *g[0].shape)), np.zeros((K, *g[1].shape)), np.zeros((K, *g[2].shape)), np.zeros(K)] for g in c.GP_prior]   # :68-79
        ref = trajectory_stats(c, state_trace[0], [v[0] for v in int_var_trace])                      # :81-93
        for i in range(c.N_int):
            for j in range(4):
                sst[i][j][0] = ref[i][j]                                                              # :94-99
        for k in range(1, K):                                                                         # :117-160
            new_state, new_int_var, _ = c(next(rands), state_trace[k - 1], [v[k - 1] for v in int_var_trace],
                                          [[sst[i][j][k - 1] for j in range(4)] for i in range(c.N_int)])   # :122-134
            state_trace[k] = np.asarray(new_state).reshape(T, nx)                                     # :137
            stats = trajectory_stats(c, state_trace[k], [np.asarray(v).reshape(T, -1) for v in new_int_var])
            for i in range(c.N_int):
                int_var_trace[i][k] = np.asarray(new_int_var[i]).reshape(T, -1)                       # :139
                for j in range(4):
                    sst[i][j][k] = stats[i][j]                                                        # :140-160
        state_trace = np.swapaxes(state_trace, 0, 1)                                                  # :161
        int_var_trace = [np.swapaxes(v, 0, 1) for v in int_var_trace]                                 # :162-165
        obs = np.stack([np.asarray(c.SSM.output_mdl(state_trace[t], c.inputs[t], *[v[t] for v in int_var_trace])).reshape(K, -1) for t in range(T)])
        ll = np.stack([c.SSM.log_likelihood(c.observations[t], state_trace[t], c.inputs[t], *[v[t] for v in int_var_trace]) for t in range(T)])
        return state_trace, int_var_trace, np.ones((T, K)) / K, sst, obs, ll                          # :180-187
