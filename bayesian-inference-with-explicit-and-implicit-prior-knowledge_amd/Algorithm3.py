"""Mirror of reference src/Algorithm3.py:16-303: the conditional (Particle-Gibbs) version of the marginalised filter, with ancestor
sampling for the reference trajectory.  Derives from Algorithm1 as the reference class does and reuses its device primitives
(include/pgas_marginal.h); see Algorithm1.py for what runs where.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .Algorithm1 import Algorithm1, STREAM_ANCESTOR, STREAM_FINAL, STREAM_RESAMPLE, _small_cholesky, _t


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
        nv = self.nvar[i]
        if nv > 1:   # the general form (BI:111-124): n M, n log det eta1, multigammaln(nu / 2, n), log det Psi
            M = P0.shape[0]
            R0 = R1 = None
            r2 = r3 = 0.0
            if ref is not None:
                R0, R1, r2, r3 = ref[0].reshape(M, nv).contiguous(), ref[1].contiguous(), ref[2].reshape(nv, nv), ref[3].reshape(())
            if sol is None:
                sol = self.ops.mniw_solve(P0, P1, T0, T1, R0=R0, R1=R1, want=("q", "logdet"))
            nu = P3 + T3 + r3
            Lp = _small_cholesky(P2 + T2 + r2 - sol["q"])                                  # Psi (N, n, n), BI:115
            logdet_psi = 2.0 * torch.log(torch.diagonal(Lp, dim1=1, dim2=2)).sum(dim=1)
            mgl = nv * (nv - 1) / 4.0 * math.log(math.pi) + sum(torch.lgamma(nu / 2 - j / 2.0) for j in range(nv))
            return (-0.5 * nv * M * math.log(2 * math.pi) + 0.5 * nv * sol["logdet"] - 0.5 * nu * nv * math.log(2.0) - mgl + logdet_psi * nu / 2)
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
        ll_aux = self.SSM.log_likelihood(self._obs(time), aux_state, self._inp(time), *aux_int_var)     # :71-84
        lw_aux = ll_aux + log_weights
        graphed = self._tidx is not None   # graph replay: the uniforms are produced on the device from the device-resident time index
        a = self.ops.systematic_resample(rand.uniform_dev(STREAM_RESAMPLE, time) if graphed else rand.uniform(STREAM_RESAMPLE, time), lw_aux.contiguous())   # :85-90
        if self.SSM.is_deterministic:
            # with process_noise == 0 (src/Toy_Example.py:66) the Gaussian of :109-116 is singular: the reference's weights are NaN and
            # its index implementation-defined; here the reference particle keeps its own ancestor
            ref_idx = N - 1
        else:
            g = None
            for i in range(self.N_int):                                                    # :93-108  g_t - g_T
                if self.nvar[i] == 1:   # both base measures and their difference in one launch, from the two solves' q / log det
                    P0, P1, P2, P3 = self.GP_prior[i]
                    T0, T1, T2, T3 = suff_stats[i]
                    R0, R1, r2, r3 = ref_suff_stats[i]
                    sol2 = self.ops.mniw_solve(P0, P1, T0, T1, R0=R0.reshape(-1).contiguous(), R1=R1.contiguous(), want=("q", "logdet"))
                    gi = self.ops.lbm_diff(P0.numel(), T2, T3, factors[i], sol2, P2, P3, r2, r3)
                else:
                    gi = self._log_base_measure(i, suff_stats[i], sol=factors[i]) - self._log_base_measure(i, suff_stats[i], ref_suff_stats[i])
                g = gi if g is None else g + gi
            if getattr(self, "_Qc", None) is None:
                Lq = np.linalg.cholesky(self.SSM.process_noise)
                self._Qc = (_t(np.linalg.inv(Lq), dev).T.contiguous(), -0.5 * Lq.shape[0] * math.log(2 * math.pi) - float(np.sum(np.log(np.diag(Lq)))))
            h_x = self.SSM.transition_logpdf(ref_state, state, self._inp(time, 1), *int_var) if hasattr(self.SSM, "transition_logpdf") else None
            if h_x is None:
                e = (ref_state.reshape(1, -1) - aux_state) @ self._Qc[0]                   # :109-116
                h_x = self._Qc[1] - 0.5 * (e * e).sum(dim=1)
            w_anc = torch.softmax(lw_aux + g + h_x, dim=0)                                 # :117-118
            u = rand.uniform_dev(STREAM_ANCESTOR, time) if graphed else torch.full((1,), rand.uniform(STREAM_ANCESTOR, time), dtype=torch.float64, device=dev)
            ref_idx = torch.clamp(torch.searchsorted(torch.cumsum(w_anc, 0), u)[0], max=N - 1)     # stays on the device: no host round trip
        a = a.clone()
        if isinstance(ref_idx, int):
            a[-1:].fill_(ref_idx)                                                          # (a fill, not a host-to-device copy: capturable)
        else:
            a[-1] = ref_idx                                                                # :121-127 (clip: SURVEY Q4)
        new_state = self._draw_states(rand, time, state, int_var, a)                       # :130-133
        new_state[-1] = ref_state                                                          # :134
        new_int_var, new_basis = self._draw_int_vars(rand, time, new_state, suff_stats, a, factors)              # :139-148
        for i in range(self.N_int):
            new_int_var[i][-1] = ref_int_var[i]                                            # :149-152
        new_stats = tuple(self.ops.stats_gather_update(1.0, a, suff_stats[i], new_basis[i], new_int_var[i].reshape(-1) if self.nvar[i] == 1 else new_int_var[i])
                          for i in range(self.N_int))                                      # :155-162
        new_ref = []
        for i in range(self.N_int):                                                        # :165-176
            rb = self.basis_fcn[i](ref_state.reshape(1, -1), self._inp(time)).reshape(-1)
            R0, R1, R2, R3 = ref_suff_stats[i]
            if self.nvar[i] > 1:
                xi = ref_int_var[i].reshape(-1)
                new_ref.append((R0.reshape(rb.numel(), -1) - rb[:, None] * xi[None, :], R1 - rb[:, None] * rb[None, :],
                                R2.reshape(xi.numel(), xi.numel()) - xi[:, None] * xi[None, :], R3.reshape(()) - 1.0))
                continue
            xi = ref_int_var[i].reshape(())
            new_ref.append((R0.reshape(-1) - rb * xi, R1 - rb[:, None] * rb[None, :], R2.reshape(()) - xi * xi, R3.reshape(()) - 1.0))
        new_lw = self.SSM.log_likelihood(self._obs(time), new_state, self._inp(time), *new_int_var) - ll_aux[a.long()]   # :179-189
        return new_lw, new_state, new_int_var, new_stats, a, tuple(new_ref)

    # ------------------------------------------------------------------------------------------------------ :199-303
    def __call__(self, key, ref_state, ref_int_var, ref_suff_stats, return_traces=False, use_graph=None):
        rand, dev = self._rand(key), self.device
        state_trace, int_var_trace, _, lw_trace, anc_trace, suff_stats = self._init_algorithm(rand)
        T = self.observations.shape[0]
        ref_state = _t(ref_state.cpu() if isinstance(ref_state, torch.Tensor) else ref_state, dev).reshape(T, -1)
        ref_int_var = [_t(v.cpu() if isinstance(v, torch.Tensor) else v, dev).reshape(T) if nv == 1 else
                       _t(v.cpu() if isinstance(v, torch.Tensor) else v, dev).reshape(T, nv) for v, nv in zip(ref_int_var, self.nvar)]
        ref_ss = [[_t(r.cpu() if isinstance(r, torch.Tensor) else r, dev) for r in rs] for rs in ref_suff_stats]
        ref_ss = [(r[0].reshape(-1), r[1], r[2].reshape(()), r[3].reshape(())) if nv == 1 else
                  (r[0].reshape(-1, nv), r[1], r[2].reshape(nv, nv), r[3].reshape(())) for r, nv in zip(ref_ss, self.nvar)]
        state_trace[0, -1] = ref_state[0]                                                  # :221
        suff_stats = [list(s) for s in suff_stats]
        for i in range(self.N_int):
            int_var_trace[i][0, -1] = ref_int_var[i][0]                                    # :224
            ib = self.basis_fcn[i](ref_state[:1], self.inputs[0]).reshape(-1)              # :225
            xi = ref_int_var[i][0]
            if self.nvar[i] > 1:
                iT = (ib[:, None] * xi[None, :], ib[:, None] * ib[None, :], xi[:, None] * xi[None, :], torch.ones((), dtype=torch.float64, device=dev))
            else:
                iT = (ib * xi, ib[:, None] * ib[None, :], xi * xi, torch.ones((), dtype=torch.float64, device=dev))   # :226
            for j in range(4):
                suff_stats[i][j][-1] = iT[j]                                               # :228-231
            ref_ss[i] = tuple(ref_ss[i][j] - iT[j] for j in range(4))                      # :235-246
        suff_stats = tuple(tuple(s) for s in suff_stats)
        def body(time, carried):                                                           # one iteration of :251-290
            stats, rss = carried
            graphed = self._tidx is not None
            prev = (lambda a: a.index_select(0, self._tidx[1]).squeeze(0)) if graphed else (lambda a: a[time - 1])
            put = (lambda a, v, back=0: a.index_copy_(0, self._tidx[back], v.reshape((1,) + a.shape[1:]).to(a.dtype))) if graphed else \
                (lambda a, v, back=0: a.__setitem__(time - back, v.reshape(a.shape[1:])))
            lw, x, iv, stats, a, rss = self.step(rand, time, prev(lw_trace), prev(state_trace), [prev(int_var_trace[i]) for i in range(self.N_int)],
                                                 stats, self._row(ref_state, time), [self._row(ref_int_var[i], time) for i in range(self.N_int)], rss)
            put(state_trace, x)
            put(lw_trace, lw)
            put(anc_trace, a, 1)
            for i in range(self.N_int):
                put(int_var_trace[i], iv[i])
            return stats, tuple(tuple(r) for r in rss)

        carried = (suff_stats, tuple(tuple(r) for r in ref_ss))
        if use_graph is None:
            use_graph = self.N_samples <= 4096 and hasattr(rand, "uniform_dev")
        if use_graph and T > 1:
            carried = self._replay(T, carried, body)                                       # Algorithm1._replay: one captured step, replayed
        else:
            for time in range(1, T):
                carried = body(time, carried)
        self.ops.check()
        w = torch.softmax(lw_trace[-1], dim=0)                                             # :293
        u = torch.tensor([rand.uniform(STREAM_FINAL, 0)], dtype=torch.float64, device=dev)
        idx = min(int(torch.searchsorted(torch.cumsum(w, 0), u)[0]), self.N_samples - 1)   # :294
        state_traj = self.ops.eng.reconstruct_trajectory(state_trace, anc_trace, idx)      # :295
        int_var_traj = tuple(self.ops.eng.reconstruct_trajectory(int_var_trace[i], anc_trace, idx) for i in range(self.N_int))   # :296-299
        if return_traces:
            return state_traj, int_var_traj, dict(state_trace=state_trace, ancestor_trace=anc_trace, log_weights=lw_trace[-1], idx=idx)
        return state_traj, int_var_traj
