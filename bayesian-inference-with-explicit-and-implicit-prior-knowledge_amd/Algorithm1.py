"""Mirror of reference src/Algorithm1.py (online marginalised particle filter); its conditional version with ancestor sampling
(src/Algorithm3.py) derives from it in Algorithm3.py.  Everything runs on the device.

Call surface = the reference's (same constructor arguments, `step`, `__call__`, same return tuples); arrays are fp64 torch
tensors on the GPU.  What runs where:

* hand-written HIP through the C ABI (include/pgas_marginal.h, include/pgas_hip.h): the per-particle MNIW algebra -- one wave
  per particle factorises eta1 = prior + statistics (M x M Cholesky in LDS) and returns the four scalars every later formula
  needs --, the ancestor gather + forgetting + rank-one statistics update, systematic resampling, all random numbers (Philox);
* torch (plumbing): the user's state-space model callables (StateSpaceModel), basis functions, and O(N) elementwise glue.

Limits: M + 1 + n <= 128 per latent function (basis size M, n components of its interface variable; n <= 8).  n = 1 with M <= 62 -- every
instantiation of the reference -- runs on the fast kernels; wider bases and n > 1 (the reference's formulas, BI:18-108, are general in n) on
the two-rows-per-lane generality kernels (Algorithm3 and Algorithm2 included).
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

    def uniform_dev(self, stream, t):
        return self.ops.uniform_dev(self.seed, stream, t)

    def student_t(self, stream, t, nu):
        return self.ops.student_t(self.seed, stream, t, nu)

    def student_t_df(self, stream, t, anc, src, nu0, nu_scale):
        return self.ops.student_t_df(self.seed, stream, t, anc, src, nu0, nu_scale)


def _small_cholesky(A):
    """Lower Cholesky factors of a batch (N, n, n) of small matrices in elementwise torch operations: no library call, no workspace, no
    host synchronisation -- the filter step that contains it can be captured in a HIP graph (torch.linalg.cholesky cannot)."""
    n = A.shape[-1]
    L = torch.zeros_like(A)
    for j in range(n):
        for i in range(j, n):
            s = A[:, i, j]
            for k in range(j):
                s = s - L[:, i, k] * L[:, j, k]
            L[:, i, j] = torch.sqrt(s) if i == j else s / L[:, j, j]
    return L


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
        if hasattr(SSM, "bind"):   # SymbolicStateSpaceModel: its traced callables run through this object's device operations
            SSM.bind(self.ops)
        self.forgetting_factor = float(forgetting_factor)
        self.init_state_mean = _t(init_state_mean, dev).reshape(-1)
        self.init_state_cov = np.atleast_2d(np.asarray(init_state_cov, dtype=np.float64))
        self.init_int_var_mean = [_t(m, dev).reshape(-1) for m in init_int_var_mean]
        self.init_int_var_cov = [np.atleast_2d(np.asarray(c, dtype=np.float64)) for c in init_int_var_cov]
        self.basis_fcn = list(basis_fcn)
        for bf in self.basis_fcn:   # descriptors evaluate the whole batch in one HIP launch once they have the device operations
            if hasattr(bf, "bind"):
                bf.bind(self.ops)
        self.N_int = len(self.basis_fcn)
        self.dim_basis = [int(self.basis_fcn[i](self.init_state_mean.reshape(1, -1), self.inputs[0]).shape[-1]) for i in range(self.N_int)]  # :56-62
        self.GP_prior, self.nvar = [], []
        for i, g in enumerate(GP_prior):
            e0, e1, e2 = np.asarray(g[0], dtype=np.float64), np.asarray(g[1], dtype=np.float64), np.atleast_2d(np.asarray(g[2], dtype=np.float64))
            M = e1.shape[0]
            nv = e0.reshape(M, -1).shape[1]
            if self.init_int_var_mean[i].numel() != nv or e2.shape != (nv, nv):
                raise ValueError(f"interface variable {i}: eta0 is (M, {nv}) but init_int_var_mean / eta2 have other sizes")
            if nv > 8 or M + 1 + nv > 128:
                raise NotImplementedError("M + 1 + n <= 128 and n <= 8 on the device path (pgas_m_mniw_solve_n: at most two matrix rows per lane)")
            self.nvar.append(nv)
            if nv == 1:   # the scalar layout of the fast kernels: eta0 (M,), eta2 a float
                self.GP_prior.append((_t(e0.reshape(-1), dev).contiguous(), _t(e1, dev).contiguous(), float(e2[0, 0]), float(g[3])))
            else:
                self.GP_prior.append((_t(e0.reshape(M, nv), dev).contiguous(), _t(e1, dev).contiguous(), _t(e2, dev).contiguous(), float(g[3])))

    # ---------------------------------------------------------------------------------------------------------------- helpers
    _tidx = None   # graph mode: (t, t - 1) as one-element int64 device tensors; rows are then gathered on the device

    def _inp(self, time, back=0):
        """inputs[time - back]; in graph mode the row is selected on the device from the time tensor."""
        if self._tidx is None:
            return self.inputs[time - back]
        return self.inputs.index_select(0, self._tidx[back]).squeeze(0)

    def _obs(self, time):
        return self.observations[time] if self._tidx is None else self.observations.index_select(0, self._tidx[0]).squeeze(0)

    def _rand(self, key):
        return key if hasattr(key, "student_t") else DeviceRand(self.ops, prng.as_key(key))

    def _ref_shapes(self, stats):
        """(T0 (N,M), T1 (N,M,M), T2 (N,), T3 (N,)) -> the reference's shapes (N,M,n), (N,M,M), (N,n,n), (N,)."""
        return tuple((s[0].reshape(s[0].shape[0], s[0].shape[1], nv), s[1], s[2].reshape(-1, nv, nv), s[3]) for s, nv in zip(stats, self.nvar))

    def _dev_shapes(self, stats):
        """n = 1: the unit axes dropped (T0 (N,M), T2 (N,)); n > 1: the reference's own shapes."""
        out = []
        for s, nv in zip(stats, self.nvar):
            N = s[0].shape[0]
            if nv == 1:
                out.append((s[0].reshape(N, -1).contiguous(), s[1].contiguous(), s[2].reshape(-1).contiguous(), s[3].reshape(-1).contiguous()))
            else:
                out.append((s[0].reshape(N, -1, nv).contiguous(), s[1].contiguous(), s[2].reshape(N, nv, nv).contiguous(), s[3].reshape(-1).contiguous()))
        return tuple(out)

    def _weighted(self, stats, w):
        """sum_n w_n T_n for the statistics trace (src/Algorithm1.py:166-170, :445-457)."""
        return self.ops.weighted_stats(w, stats)

    # ------------------------------------------------------------------------------------------------------ :100-177
    def _init_trace_vars(self):
        T, N, dev = self.observations.shape[0], self.N_samples, self.device
        z = lambda *s: torch.zeros(s, dtype=torch.float64, device=dev)  # noqa: E731
        return (z(T, N, self.init_state_mean.numel()), [z(T, N, nv) for nv in self.nvar],
                [[z(T, M, nv), z(T, M, M), z(T, nv, nv), z(T)] for M, nv in zip(self.dim_basis, self.nvar)], z(T, N),
                torch.zeros((T - 1, N), dtype=torch.int32, device=dev))

    def _init_algorithm(self, rand):
        state_trace, int_var_trace, sst, lw_trace, anc_trace = self._init_trace_vars()
        dev, N = self.device, self.N_samples
        nx = self.init_state_mean.numel()
        state_trace[0] = self.init_state_mean + rand.normal(STREAM_INIT_STATE, 0, nx) @ _t(np.linalg.cholesky(self.init_state_cov), dev).T  # :139-145
        suff_stats = []
        w = torch.full((N,), 1.0 / N, dtype=torch.float64, device=dev)                     # softmax of zeros, :166
        for i in range(self.N_int):
            nv = self.nvar[i]
            basis = self.basis_fcn[i](state_trace[0], self.inputs[0]).contiguous()         # :158-160
            if nv == 1:
                sd = float(np.sqrt(self.init_int_var_cov[i][0, 0]))
                int_var_trace[i][0] = self.init_int_var_mean[i] + sd * rand.normal(STREAM_INIT_INTVAR + i, 0, 1)      # :146-153
                xi = int_var_trace[i][0].reshape(-1)
                Ts = ((basis * xi[:, None]).contiguous(), (basis[:, :, None] * basis[:, None, :]).contiguous(), xi * xi, torch.ones_like(xi))   # :161-163
            else:
                int_var_trace[i][0] = self.init_int_var_mean[i] + rand.normal(STREAM_INIT_INTVAR + i, 0, nv) @ _t(np.linalg.cholesky(self.init_int_var_cov[i]), dev).T
                xi = int_var_trace[i][0]
                Ts = ((basis[:, :, None] * xi[:, None, :]).contiguous(), (basis[:, :, None] * basis[:, None, :]).contiguous(),
                      (xi[:, :, None] * xi[:, None, :]).contiguous(), torch.ones(N, dtype=torch.float64, device=dev))
            suff_stats.append(Ts)
            for j, v in enumerate(self._weighted(Ts, w)):
                sst[i][j][0] = v.reshape(sst[i][j][0].shape)                               # :167-170
        return state_trace, int_var_trace, sst, lw_trace, anc_trace, tuple(suff_stats)

    # ------------------------------------------------------------------------------------------------------ :179-232
    def _generate_auxiliary_states(self, state, time, int_var, suff_stats, scale=1.0):
        """Returns (aux_state, aux_int_var, factors).  factors[i] keeps, per particle, the Cholesky factor of eta1 = prior + scale T1,
        w = L^-1 eta0, q = w . w and log det eta1: the resampled children reuse them in _draw_int_vars (their matrix is their
        ancestor's, :358-361) and Algorithm3 reads q / logdet for its base measures -- one factorisation per particle and step."""
        aux_state = self.SSM.transition_mdl(state, self._inp(time, 1), *int_var)        # :206-208
        aux_int_var, factors = [], []
        for i in range(self.N_int):
            basis = self.basis_fcn[i](aux_state, self._inp(time)).contiguous()              # :220-225
            P0, P1, _, _ = self.GP_prior[i]
            # mean_i phi = eta0^T eta1^-1 phi (BI:48-50, :228-231); `scale` carries the forgetting factor of :317-320
            sol = self.ops.mniw_solve(P0, P1, suff_stats[i][0], suff_stats[i][1], scale=scale, phi=basis, want=("m", "q", "logdet"), keep_factor=True)
            aux_int_var.append(sol["m"].unsqueeze(-1) if self.nvar[i] == 1 else sol["m"])   # (N, n)
            factors.append(sol)
        return aux_state, tuple(aux_int_var), factors

    # ------------------------------------------------------------------------------------------------------ :234-273
    def _draw_int_vars(self, rand, time, state, suff_stats, a, factors, scale=1.0):
        """Interface variables drawn from the matrix-t predictive of the RESAMPLED statistics suff_stats[.][a] (:251-262): a triangular
        solve against the ancestor's stored factor.  Returns (int_var, basis)."""
        int_var, basis_all = [], []
        ai = a.long()
        for i in range(self.N_int):
            basis = self.basis_fcn[i](state, self._inp(time)).contiguous()                  # :243-248
            _, _, P2, P3 = self.GP_prior[i]
            _, _, T2, T3 = suff_stats[i]
            sol = self.ops.mniw_trisolve(factors[i], a, basis)                             # m = mean phi (BI:81), c = phi^T col_cov phi (BI:84)
            if self.nvar[i] > 1:
                nv = self.nvar[i]
                df = P3 + scale * T3[ai] + 1.0 - nv                                        # BI:45, :78
                row = (P2 + scale * T2[ai] - factors[i]["q"][ai]) / df[:, None, None]       # BI:42, :87 -- (N, n, n)
                Lr = _small_cholesky(row)                                                  # BI:100
                t = torch.stack([rand.student_t(STREAM_INTVAR + i + 16 * j, time, df) for j in range(nv)], dim=1)   # BI:104: n variates per particle
                draw = sol["m"] + torch.einsum("nij,nj->ni", Lr, t) * torch.sqrt(sol["c"] + 1.0)[:, None]           # BI:106-108 (col_scale is 1 x 1)
                int_var.append(draw)
                basis_all.append(basis)
                continue
            if hasattr(rand, "student_t_df"):   # device random numbers: degrees of freedom and the draw formed inside two kernels
                t = rand.student_t_df(STREAM_INTVAR + i, time, a, T3, P3, scale)           # BI:45, :78, :104
                int_var.append(self.ops.mniw_draw(scale, a, sol["m"], sol["c"], factors[i]["q"], T2, T3, P2, P3, t).unsqueeze(-1))   # BI:81-108
                basis_all.append(basis)
                continue
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
        z = rand.normal(STREAM_STATE, time, state.shape[1])
        if hasattr(self.SSM, "draw_state_gather"):   # the traced model gathers its parents' rows itself
            return self.SSM.draw_state_gather(z, state, self._inp(time, 1), a, *int_var)
        ai = a.long()
        return self.SSM.draw_state(z, state[ai], self._inp(time, 1), *[v[ai] for v in int_var])

    # ------------------------------------------------------------------------------------------------------ :297-397
    def step(self, key, time, log_weights, state, int_var, suff_stats):
        """Returns (new_log_weights (N,), new_state (N,n_x), new_int_var, new_suff_stats, a_indices (N,) int32).  `suff_stats` in and
        out: per interface variable (T0 (N,M), T1 (N,M,M), T2 (N,), T3 (N,)) -- the reference's arrays with the unit axes dropped."""
        rand, time, lam = self._rand(key), int(time), self.forgetting_factor
        suff_stats = self._dev_shapes(suff_stats)
        # :317-320 statistics time update: the factor is applied inside the kernels (scale * T), never materialised
        aux_state, aux_int_var, factors = self._generate_auxiliary_states(state, time, int_var, suff_stats, scale=lam)   # :323-325
        ll_aux = self.SSM.log_likelihood(self._obs(time), aux_state, self._inp(time), *aux_int_var)    # :328-341
        u = rand.uniform(STREAM_RESAMPLE, time) if self._tidx is None else rand.uniform_dev(STREAM_RESAMPLE, time)
        a = self.ops.systematic_resample(u, (ll_aux + log_weights).contiguous())       # :342-347
        new_state = self._draw_states(rand, time, state, int_var, a)                       # :350-353
        new_int_var, new_basis = self._draw_int_vars(rand, time, new_state, suff_stats, a, factors, scale=lam)   # :358-367
        new_stats = tuple(self.ops.stats_gather_update(lam, a, suff_stats[i], new_basis[i], new_int_var[i].reshape(-1) if self.nvar[i] == 1 else new_int_var[i])
                          for i in range(self.N_int))                                      # :370-377
        new_lw = self.SSM.log_likelihood(self._obs(time), new_state, self._inp(time), *new_int_var) - ll_aux[a.long()]   # :380-390
        return new_lw, new_state, new_int_var, new_stats, a

    # ------------------------------------------------------------------------------------------------------ :399-492
    def _loop_body(self, rand, time, traces, suff_stats):
        """One iteration of the reference loop (:418-457) against the trace arrays; returns the new per-particle statistics."""
        state_trace, int_var_trace, sst, lw_trace, anc_trace = traces
        if self._tidx is None:
            prev = lambda a: a[time - 1]                                                   # noqa: E731
            put = lambda a, v, back=0: a.__setitem__(time - back, v.reshape(a.shape[1:]))   # noqa: E731
        else:   # graph mode: rows addressed through the device-resident time index
            prev = lambda a: a.index_select(0, self._tidx[1]).squeeze(0)                   # noqa: E731
            put = lambda a, v, back=0: a.index_copy_(0, self._tidx[back], v.reshape((1,) + a.shape[1:]).to(a.dtype))   # noqa: E731
        lw, x, iv, suff_stats, a = self.step(rand, time, prev(lw_trace), prev(state_trace), [prev(int_var_trace[i]) for i in range(self.N_int)], suff_stats)
        put(state_trace, x)
        put(lw_trace, lw)
        put(anc_trace, a, 1)
        w = torch.softmax(lw, dim=0)
        for i in range(self.N_int):
            put(int_var_trace[i], iv[i])
            for j, v in enumerate(self._weighted(suff_stats[i], w)):
                put(sst[i][j], v)                                                          # :445-457
        return suff_stats

    def _graphed_loop(self, rand, traces, suff_stats, T):
        return self._replay(T, suff_stats, lambda time, carried: self._loop_body(rand, time, traces, carried))

    def _row(self, a, time):
        """a[time]; in graph mode the row is selected on the device from the time tensor."""
        return a[time] if self._tidx is None else a.index_select(0, self._tidx[0]).squeeze(0)

    def _replay(self, T, carried, body):
        """Run `carried = body(time, carried)` for time = 1 .. T-1 as ONE captured HIP graph replayed for t = 3 .. T-1: a step is ~100
        small launches, which at the reference's N = 200 is pure launch latency.  The time index lives on the device (random-number
        counters: pgas_m_set_time_source; rows of inputs / observations / traces: index_select / index_copy_ on it) and is incremented
        inside the graph; `carried` (a nested tuple of tensors: the per-particle statistics, ...) lives in static buffers.  Same kernels
        in the same order as the eager loop: identical results."""
        dev = self.device

        def flat(x):
            return [x] if isinstance(x, torch.Tensor) else [t for y in x for t in flat(y)]

        def clone(x):
            return x.clone() if isinstance(x, torch.Tensor) else tuple(clone(y) for y in x)

        carried = body(1, carried)                                         # t = 1 eagerly: lazy allocations, library warm-up
        if T <= 2:
            return carried
        t32 = torch.full((1,), 2, dtype=torch.int32, device=dev)          # what the random-number kernels read
        self._tidx = (torch.full((1,), 2, dtype=torch.int64, device=dev), torch.full((1,), 1, dtype=torch.int64, device=dev))
        carried = clone(carried)
        self.ops.set_time_source(t32)
        try:
            def one():
                new = body(0, carried)                                     # `time` is ignored in graph mode
                for dst, src in zip(flat(carried), flat(new)):
                    dst.copy_(src.reshape(dst.shape))
                t32.add_(1)
                self._tidx[0].add_(1)
                self._tidx[1].add_(1)

            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                one()                                                      # t = 2 eagerly on a side stream (capture warm-up)
            torch.cuda.current_stream(dev).wait_stream(side)
            if T > 3:
                graph = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(graph):
                        one()
                except Exception as e:   # noqa: BLE001
                    # A model callable that synchronises with the host (.item(), .cpu(), a Python scalar assigned into a tensor) cannot be
                    # captured.  Nothing of the failed capture has executed and the carried buffers / device time index stand at t = 3,
                    # so the same step body simply goes on launch by launch (INTEGRATION.md lists what a captured callable may do).
                    import warnings

                    warnings.warn(f"the filter step could not be captured in a HIP graph ({type(e).__name__}: {e}); continuing without graph "
                                  "replay -- pass use_graph=False to skip the attempt", RuntimeWarning, stacklevel=3)
                    graph = None
                    try:   # some failures (an operation HIP refuses inside a capture) invalidate the stream for good: say so instead of failing later, somewhere else
                        torch.cuda.current_stream(dev).synchronize()
                        (torch.zeros(1, device=dev) + 1).item()
                    except Exception as e2:   # noqa: BLE001
                        raise RuntimeError("the failed graph capture left the HIP stream unusable (" + type(e2).__name__ + "); this process cannot continue on "
                                           "the device -- start again with use_graph=False") from e
                for _ in range(3, T):                                      # the capture itself does not execute
                    if graph is not None:
                        graph.replay()
                    else:
                        one()
            torch.cuda.current_stream(dev).synchronize()
        finally:
            self.ops.set_time_source(None)
            self._tidx = None
        return carried

    def __call__(self, key, use_graph=None):
        """use_graph: capture the filter step in a HIP graph and replay it (default: when N_samples <= 4096, the launch-latency-bound
        regime; the statistics are then copied once per step, which costs nothing there and 14 GB of traffic per step at N = 2^20)."""
        rand = self._rand(key)
        state_trace, int_var_trace, sst, lw_trace, anc_trace, suff_stats = self._init_algorithm(rand)
        T = self.observations.shape[0]
        traces = (state_trace, int_var_trace, sst, lw_trace, anc_trace)
        if use_graph is None:
            use_graph = self.N_samples <= 4096 and isinstance(rand, DeviceRand)
        if use_graph and T > 1:
            suff_stats = self._graphed_loop(rand, traces, suff_stats, T)
        else:
            for time in range(1, T):
                suff_stats = self._loop_body(rand, time, traces, suff_stats)
        self.ops.check()
        weights_trace = torch.softmax(lw_trace, dim=1)                                     # :460
        obs_trace = torch.stack([self.SSM.output_mdl(state_trace[t], self.inputs[t], *[v[t] for v in int_var_trace]).reshape(self.N_samples, -1)
                                 for t in range(T)])                                       # :463-468
        loglik = torch.stack([self.SSM.log_likelihood(self.observations[t], state_trace[t], self.inputs[t], *[v[t] for v in int_var_trace])
                              for t in range(T)])                                          # :471-481
        return state_trace, int_var_trace, sst, weights_trace, anc_trace, self._ref_shapes(suff_stats), obs_trace, loglik
