"""Mirror of reference src/Algorithm1.py (online marginalised particle filter) and src/Algorithm3.py (its conditional version
with ancestor sampling) on the device.

Call surface = the reference's (same constructor arguments, `step`, `__call__`, same return tuples); arrays are fp64 torch
tensors on the GPU.  What runs where:

* hand-written HIP through the C ABI (include/pgas_marginal.h, include/pgas_hip.h): the per-particle MNIW algebra -- one wave
  per particle factorises eta1 = prior + statistics (M x M Cholesky in LDS) and returns the four scalars every later formula
  needs --, the ancestor gather + forgetting + rank-one statistics update, systematic resampling, all random numbers (Philox);
* torch (plumbing): the user's state-space model callables (StateSpaceModel), basis functions, and O(N) elementwise glue.

Restrictions: scalar interface variables (n = 1, as in every instantiation of the reference); basis size M <= 62.
`key` is an integer seed (own Philox streams, include/pgas_canon.h) or an object with the provider interface of `DeviceRand`.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import random as prng
from ._lib import MarginalOps
from .StateSpaceModel import StateSpaceModel  # noqa: F401  (re-exported like the reference module does)

STREAM_INIT_STATE, STREAM_STATE, STREAM_RESAMPLE, STREAM_ANCESTOR, STREAM_FINAL, STREAM_INIT_INTVAR, STREAM_INTVAR = 16, 17, 18, 19, 20, 24, 32


class DeviceRand:
    """Random numbers of one run: Philox streams addressed by (stream, time step, particle), generated on the device."""

    def __init__(self, ops: MarginalOps, seed: int):
        self.ops, self.seed = ops, int(seed)

    def normal(self, stream, t, ncol):
        return self.ops.normal(self.seed, stream, t, ncol)

    def uniform(self, stream, t):
        return self.ops.uniform(self.seed, stream, t)

    def student_t(self, stream, t, nu):
        return self.ops.student_t(self.seed, stream, t, nu)


def _t(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)


class Algorithm1:
    def __init__(self, N_samples, observations, inputs, SSM, forgetting_factor, init_state_mean, init_state_cov, init_int_var_mean,
                 init_int_var_cov, GP_prior, basis_fcn, device=None):
        self.N_samples = int(N_samples)
        self.ops = MarginalOps(self.N_samples, device)
        dev = self.device = self.ops.device
        self.observations = _t(observations, dev)
        self.inputs = _t(inputs, dev)
        self.SSM = SSM
        self.forgetting_factor = float(forgetting_factor)
        self.init_state_mean = _t(init_state_mean, dev).reshape(-1)
        self.init_state_cov = np.atleast_2d(np.asarray(init_state_cov, dtype=np.float64))
        self.init_int_var_mean = [_t(m, dev).reshape(-1) for m in init_int_var_mean]
        self.init_int_var_cov = [np.atleast_2d(np.asarray(c, dtype=np.float64)) for c in init_int_var_cov]
        self.basis_fcn = list(basis_fcn)
        self.N_int = len(self.basis_fcn)
        self.dim_basis = [int(self.basis_fcn[i](self.init_state_mean.reshape(1, -1), self.inputs[0]).shape[-1]) for i in range(self.N_int)]  # :56-62
        self.GP_prior = []
        for i, g in enumerate(GP_prior):
            e0, e1, e2 = np.asarray(g[0], dtype=np.float64), np.asarray(g[1], dtype=np.float64), np.atleast_2d(np.asarray(g[2], dtype=np.float64))
            if e0.reshape(e1.shape[0], -1).shape[1] != 1 or self.init_int_var_mean[i].numel() != 1:
                raise NotImplementedError("the device path handles scalar interface variables (n = 1), as every reference configuration has")
            if e1.shape[0] > 62:
                raise NotImplementedError("basis size M <= 62 on the device path (pgas_m_mniw_solve: M + 2 rows, one per lane)")
            self.GP_prior.append((_t(e0.reshape(-1), dev).contiguous(), _t(e1, dev).contiguous(), float(e2[0, 0]), float(g[3])))

    # ---------------------------------------------------------------------------------------------------------------- helpers
    def _rand(self, key):
        return key if hasattr(key, "student_t") else DeviceRand(self.ops, prng.as_key(key))

    @staticmethod
    def _ref_shapes(stats):
        """(T0 (N,M), T1 (N,M,M), T2 (N,), T3 (N,)) -> the reference's shapes (N,M,1), (N,M,M), (N,1,1), (N,)."""
        return tuple((s[0].unsqueeze(-1), s[1], s[2].reshape(-1, 1, 1), s[3]) for s in stats)

    @staticmethod
    def _dev_shapes(stats):
        return tuple((s[0].reshape(s[0].shape[0], -1).contiguous(), s[1].contiguous(), s[2].reshape(-1).contiguous(), s[3].reshape(-1).contiguous()) for s in stats)

    def _weighted(self, stats, w):
        """sum_n w_n T_n for the statistics trace (src/Algorithm1.py:166-170, :445-457)."""
        return self.ops.weighted_stats(w, stats)

    # ------------------------------------------------------------------------------------------------------ :100-177
    def _init_trace_vars(self):
        T, N, dev = self.observations.shape[0], self.N_samples, self.device
        z = lambda *s: torch.zeros(s, dtype=torch.float64, device=dev)  # noqa: E731
        return (z(T, N, self.init_state_mean.numel()), [z(T, N, 1) for _ in range(self.N_int)],
                [[z(T, M, 1), z(T, M, M), z(T, 1, 1), z(T)] for M in self.dim_basis], z(T, N),
                torch.zeros((T - 1, N), dtype=torch.int32, device=dev))

    def _init_algorithm(self, rand):
        state_trace, int_var_trace, sst, lw_trace, anc_trace = self._init_trace_vars()
        dev, N = self.device, self.N_samples
        nx = self.init_state_mean.numel()
        state_trace[0] = self.init_state_mean + rand.normal(STREAM_INIT_STATE, 0, nx) @ _t(np.linalg.cholesky(self.init_state_cov), dev).T  # :139-145
        suff_stats = []
        w = torch.full((N,), 1.0 / N, dtype=torch.float64, device=dev)                     # softmax of zeros, :166
        for i in range(self.N_int):
            sd = float(np.sqrt(self.init_int_var_cov[i][0, 0]))
            int_var_trace[i][0] = self.init_int_var_mean[i] + sd * rand.normal(STREAM_INIT_INTVAR + i, 0, 1)      # :146-153
            basis = self.basis_fcn[i](state_trace[0], self.inputs[0]).contiguous()         # :158-160
            xi = int_var_trace[i][0].reshape(-1)
            Ts = ((basis * xi[:, None]).contiguous(), (basis[:, :, None] * basis[:, None, :]).contiguous(), xi * xi, torch.ones_like(xi))   # :161-163
            suff_stats.append(Ts)
            for j, v in enumerate(self._weighted(Ts, w)):
                sst[i][j][0] = v.reshape(sst[i][j][0].shape)                               # :167-170
        return state_trace, int_var_trace, sst, lw_trace, anc_trace, tuple(suff_stats)

    # ------------------------------------------------------------------------------------------------------ :179-232
    def _generate_auxiliary_states(self, state, time, int_var, suff_stats, scale=1.0):
        """Returns (aux_state, aux_int_var, factors).  factors[i] keeps, per particle, the Cholesky factor of eta1 = prior + scale T1,
        w = L^-1 eta0, q = w . w and log det eta1: the resampled children reuse them in _draw_int_vars (their matrix is their
        ancestor's, :358-361) and Algorithm3 reads q / logdet for its base measures -- one factorisation per particle and step."""
        aux_state = self.SSM.transition_mdl(state, self.inputs[time - 1], *int_var)        # :206-208
        aux_int_var, factors = [], []
        for i in range(self.N_int):
            basis = self.basis_fcn[i](aux_state, self.inputs[time]).contiguous()           # :220-225
            P0, P1, _, _ = self.GP_prior[i]
            # mean_i phi = eta0^T eta1^-1 phi (BI:48-50, :228-231); `scale` carries the forgetting factor of :317-320
            sol = self.ops.mniw_solve(P0, P1, suff_stats[i][0], suff_stats[i][1], scale=scale, phi=basis, want=("m", "q", "logdet"), keep_factor=True)
            aux_int_var.append(sol["m"].unsqueeze(-1))
            factors.append(sol)
        return aux_state, tuple(aux_int_var), factors

    # ------------------------------------------------------------------------------------------------------ :234-273
    def _draw_int_vars(self, rand, time, state, suff_stats, a, factors, scale=1.0):
        """Interface variables drawn from the matrix-t predictive of the RESAMPLED statistics suff_stats[.][a] (:251-262): a triangular
        solve against the ancestor's stored factor.  Returns (int_var, basis)."""
        int_var, basis_all = [], []
        ai = a.long()
        for i in range(self.N_int):
            basis = self.basis_fcn[i](state, self.inputs[time]).contiguous()               # :243-248
            _, _, P2, P3 = self.GP_prior[i]
            _, _, T2, T3 = suff_stats[i]
            sol = self.ops.mniw_trisolve(factors[i], a, basis)                             # m = mean phi (BI:81), c = phi^T col_cov phi (BI:84)
            df = P3 + scale * T3[ai]                                                       # BI:45; BI:78 with n = 1: df + 1 - 1
            row_scale = (P2 + scale * T2[ai] - factors[i]["q"][ai]) / df                   # BI:42, :87
            col_scale = sol["c"] + 1.0                                                     # BI:84
            t = rand.student_t(STREAM_INTVAR + i, time, df)                                # BI:104
            draw = sol["m"] + torch.sqrt(row_scale) * t * torch.sqrt(col_scale)            # BI:96-108 (Cholesky of 1 x 1 matrices)
            int_var.append(draw.unsqueeze(-1))
            basis_all.append(basis)
        return tuple(int_var), tuple(basis_all)

    # ------------------------------------------------------------------------------------------------------ :275-295
    def _draw_states(self, rand, time, state, int_var, a):
        ai = a.long()
        z = rand.normal(STREAM_STATE, time, state.shape[1])
        return self.SSM.draw_state(z, state[ai], self.inputs[time - 1], *[v[ai] for v in int_var])

    # ------------------------------------------------------------------------------------------------------ :297-397
    def step(self, key, time, log_weights, state, int_var, suff_stats):
        """Returns (new_log_weights (N,), new_state (N,n_x), new_int_var, new_suff_stats, a_indices (N,) int32).  `suff_stats` in and
        out: per interface variable (T0 (N,M), T1 (N,M,M), T2 (N,), T3 (N,)) -- the reference's arrays with the unit axes dropped."""
        rand, time, lam = self._rand(key), int(time), self.forgetting_factor
        suff_stats = self._dev_shapes(suff_stats)
        # :317-320 statistics time update: the factor is applied inside the kernels (scale * T), never materialised
        aux_state, aux_int_var, factors = self._generate_auxiliary_states(state, time, int_var, suff_stats, scale=lam)   # :323-325
        ll_aux = self.SSM.log_likelihood(self.observations[time], aux_state, self.inputs[time], *aux_int_var)    # :328-341
        a = self.ops.systematic_resample(rand.uniform(STREAM_RESAMPLE, time), (ll_aux + log_weights).contiguous())   # :342-347
        new_state = self._draw_states(rand, time, state, int_var, a)                       # :350-353
        new_int_var, new_basis = self._draw_int_vars(rand, time, new_state, suff_stats, a, factors, scale=lam)   # :358-367
        new_stats = tuple(self.ops.stats_gather_update(lam, a, suff_stats[i], new_basis[i], new_int_var[i].reshape(-1))
                          for i in range(self.N_int))                                      # :370-377
        new_lw = self.SSM.log_likelihood(self.observations[time], new_state, self.inputs[time], *new_int_var) - ll_aux[a.long()]   # :380-390
        return new_lw, new_state, new_int_var, new_stats, a

    # ------------------------------------------------------------------------------------------------------ :399-492
    def __call__(self, key):
        rand = self._rand(key)
        state_trace, int_var_trace, sst, lw_trace, anc_trace, suff_stats = self._init_algorithm(rand)
        T = self.observations.shape[0]
        for time in range(1, T):
            lw, x, iv, suff_stats, a = self.step(rand, time, lw_trace[time - 1], state_trace[time - 1],
                                                 [int_var_trace[i][time - 1] for i in range(self.N_int)], suff_stats)
            state_trace[time], lw_trace[time], anc_trace[time - 1] = x, lw, a
            w = torch.softmax(lw, dim=0)
            for i in range(self.N_int):
                int_var_trace[i][time] = iv[i]
                for j, v in enumerate(self._weighted(suff_stats[i], w)):
                    sst[i][j][time] = v.reshape(sst[i][j][time].shape)                     # :445-457
        self.ops.check()
        weights_trace = torch.softmax(lw_trace, dim=1)                                     # :460
        obs_trace = torch.stack([self.SSM.output_mdl(state_trace[t], self.inputs[t], *[v[t] for v in int_var_trace]).reshape(self.N_samples, -1)
                                 for t in range(T)])                                       # :463-468
        loglik = torch.stack([self.SSM.log_likelihood(self.observations[t], state_trace[t], self.inputs[t], *[v[t] for v in int_var_trace])
                              for t in range(T)])                                          # :471-481
        return state_trace, int_var_trace, sst, weights_trace, anc_trace, self._ref_shapes(suff_stats), obs_trace, loglik


class Algorithm3(Algorithm1):
    """src/Algorithm3.py:15-303 (forgetting factor fixed to 1.0 and never applied)."""

    def __init__(self, N_samples, observations, inputs, SSM, init_state_mean, init_state_cov, init_int_var_mean, init_int_var_cov, GP_prior,
                 basis_fcn, device=None):
        super().__init__(N_samples, observations, inputs, SSM, 1.0, init_state_mean, init_state_cov, init_int_var_mean, init_int_var_cov,
                         GP_prior, basis_fcn, device=device)

    def _log_base_measure(self, i, stats, ref=None, sol=None):
        """vmap(BI.prior_mniw_log_base_measure) (BI:111-124) of prior + stats (+ ref) for n = 1: multigammaln(a, 1) = lgamma(a).
        `sol` = (q, logdet) already computed for the same matrices (the auxiliary pass of this step)."""
        P0, P1, P2, P3 = self.GP_prior[i]
        T0, T1, T2, T3 = stats
        M = P0.numel()
        R0 = R1 = None
        r2 = r3 = 0.0
        if ref is not None:
            R0, R1, r2, r3 = ref[0].reshape(-1).contiguous(), ref[1].contiguous(), ref[2].reshape(()), ref[3].reshape(())
        if sol is None:
            sol = self.ops.mniw_solve(P0, P1, T0, T1, R0=R0, R1=R1, want=("q", "logdet"))
        nu = P3 + T3 + r3
        Psi = P2 + T2 + r2 - sol["q"]                                                      # BI:115
        return (-0.5 * M * math.log(2 * math.pi) + 0.5 * sol["logdet"] - 0.5 * nu * math.log(2.0) - torch.lgamma(nu / 2)
                + torch.log(Psi) * nu / 2)                                                 # BI:118-124

    # ------------------------------------------------------------------------------------------------------ :43-197
    def step(self, key, time, log_weights, state, int_var, suff_stats, ref_state, ref_int_var, ref_suff_stats):
        """ref_suff_stats: per interface variable (T0 (M,[1]), T1 (M,M), T2, T3) of the remaining reference trajectory."""
        rand, time, N, dev = self._rand(key), int(time), self.N_samples, self.device
        suff_stats = self._dev_shapes(suff_stats)
        ref_state = _t(ref_state, dev).reshape(-1) if not isinstance(ref_state, torch.Tensor) else ref_state.reshape(-1)
        ref_int_var = [(_t(v, dev) if not isinstance(v, torch.Tensor) else v).reshape(-1) for v in ref_int_var]
        ref_suff_stats = [tuple((_t(r, dev) if not isinstance(r, torch.Tensor) else r) for r in rs) for rs in ref_suff_stats]
        aux_state, aux_int_var, factors = self._generate_auxiliary_states(state, time, int_var, suff_stats)      # :66-68
        ll_aux = self.SSM.log_likelihood(self.observations[time], aux_state, self.inputs[time], *aux_int_var)     # :71-84
        lw_aux = ll_aux + log_weights
        a = self.ops.systematic_resample(rand.uniform(STREAM_RESAMPLE, time), lw_aux.contiguous())                # :85-90
        if self.SSM.is_deterministic:
            # with process_noise == 0 (src/Toy_Example.py:66) the Gaussian of :109-116 is singular: the reference's weights are NaN and
            # its index implementation-defined; here the reference particle keeps its own ancestor
            ref_idx = N - 1
        else:
            g = torch.zeros(N, dtype=torch.float64, device=dev)
            for i in range(self.N_int):                                                    # :93-108  g_t - g_T
                g = g + self._log_base_measure(i, suff_stats[i], sol=factors[i]) - self._log_base_measure(i, suff_stats[i], ref_suff_stats[i])
            if getattr(self, "_Qc", None) is None:
                Lq = np.linalg.cholesky(self.SSM.process_noise)
                self._Qc = (_t(np.linalg.inv(Lq), dev).T.contiguous(), -0.5 * Lq.shape[0] * math.log(2 * math.pi) - float(np.sum(np.log(np.diag(Lq)))))
            e = (ref_state.reshape(1, -1) - aux_state) @ self._Qc[0]                       # :109-116
            h_x = self._Qc[1] - 0.5 * (e * e).sum(dim=1)
            w_anc = torch.softmax(lw_aux + g + h_x, dim=0)                                 # :117-118
            u = torch.full((1,), rand.uniform(STREAM_ANCESTOR, time), dtype=torch.float64, device=dev)
            ref_idx = torch.clamp(torch.searchsorted(torch.cumsum(w_anc, 0), u)[0], max=N - 1)     # stays on the device: no host round trip
        a = a.clone()
        a[-1] = ref_idx                                                                    # :121-127 (clip: SURVEY Q4)
        new_state = self._draw_states(rand, time, state, int_var, a)                       # :130-133
        new_state[-1] = ref_state                                                          # :134
        new_int_var, new_basis = self._draw_int_vars(rand, time, new_state, suff_stats, a, factors)              # :139-148
        for i in range(self.N_int):
            new_int_var[i][-1] = ref_int_var[i]                                            # :149-152
        new_stats = tuple(self.ops.stats_gather_update(1.0, a, suff_stats[i], new_basis[i], new_int_var[i].reshape(-1))
                          for i in range(self.N_int))                                      # :155-162
        new_ref = []
        for i in range(self.N_int):                                                        # :165-176
            rb = self.basis_fcn[i](ref_state.reshape(1, -1), self.inputs[time]).reshape(-1)
            xi = ref_int_var[i].reshape(())
            R0, R1, R2, R3 = ref_suff_stats[i]
            new_ref.append((R0.reshape(-1) - rb * xi, R1 - rb[:, None] * rb[None, :], R2.reshape(()) - xi * xi, R3.reshape(()) - 1.0))
        new_lw = self.SSM.log_likelihood(self.observations[time], new_state, self.inputs[time], *new_int_var) - ll_aux[a.long()]   # :179-189
        return new_lw, new_state, new_int_var, new_stats, a, tuple(new_ref)

    # ------------------------------------------------------------------------------------------------------ :199-303
    def __call__(self, key, ref_state, ref_int_var, ref_suff_stats, return_traces=False):
        rand, dev = self._rand(key), self.device
        state_trace, int_var_trace, _, lw_trace, anc_trace, suff_stats = self._init_algorithm(rand)
        T = self.observations.shape[0]
        ref_state = _t(ref_state.cpu() if isinstance(ref_state, torch.Tensor) else ref_state, dev).reshape(T, -1)
        ref_int_var = [_t(v.cpu() if isinstance(v, torch.Tensor) else v, dev).reshape(T) for v in ref_int_var]
        ref_ss = [[_t(r.cpu() if isinstance(r, torch.Tensor) else r, dev) for r in rs] for rs in ref_suff_stats]
        ref_ss = [(r[0].reshape(-1), r[1], r[2].reshape(()), r[3].reshape(())) for r in ref_ss]
        state_trace[0, -1] = ref_state[0]                                                  # :221
        suff_stats = [list(s) for s in suff_stats]
        for i in range(self.N_int):
            int_var_trace[i][0, -1] = ref_int_var[i][0]                                    # :224
            ib = self.basis_fcn[i](ref_state[:1], self.inputs[0]).reshape(-1)              # :225
            xi = ref_int_var[i][0]
            iT = (ib * xi, ib[:, None] * ib[None, :], xi * xi, torch.ones((), dtype=torch.float64, device=dev))   # :226
            for j in range(4):
                suff_stats[i][j][-1] = iT[j]                                               # :228-231
            ref_ss[i] = tuple(ref_ss[i][j] - iT[j] for j in range(4))                      # :235-246
        suff_stats = tuple(tuple(s) for s in suff_stats)
        for time in range(1, T):                                                           # :251-290
            lw, x, iv, suff_stats, a, ref_ss = self.step(rand, time, lw_trace[time - 1], state_trace[time - 1],
                                                         [int_var_trace[i][time - 1] for i in range(self.N_int)], suff_stats,
                                                         ref_state[time], [ref_int_var[i][time] for i in range(self.N_int)], ref_ss)
            state_trace[time], lw_trace[time], anc_trace[time - 1] = x, lw, a
            for i in range(self.N_int):
                int_var_trace[i][time] = iv[i]
        self.ops.check()
        w = torch.softmax(lw_trace[-1], dim=0)                                             # :293
        u = torch.tensor([rand.uniform(STREAM_FINAL, 0)], dtype=torch.float64, device=dev)
        idx = min(int(torch.searchsorted(torch.cumsum(w, 0), u)[0]), self.N_samples - 1)   # :294
        state_traj = self.ops.eng.reconstruct_trajectory(state_trace, anc_trace, idx)      # :295
        int_var_traj = tuple(self.ops.eng.reconstruct_trajectory(int_var_trace[i], anc_trace, idx) for i in range(self.N_int))   # :296-299
        if return_traces:
            return state_traj, int_var_traj, dict(state_trace=state_trace, ancestor_trace=anc_trace, log_weights=lw_trace[-1], idx=idx)
        return state_traj, int_var_traj
