"""Mirror of reference src/Algorithm2.py:12-187: Particle Gibbs over the marginalised conditional filter (Algorithm3).

Same constructor and `__call__(key, init_ref_state, init_ref_int_var)` as the reference, same 6-tuple
(state_trace (T,K,n_x), int_var_trace [(T,K,1)], weights ones/K (T,K), suff_stats_trace [[(K,M,1),(K,M,M),(K,1,1),(K,)]],
obs_trace (T,K,n_y), log_likelihood (T,K)); arrays are fp64 torch tensors on the device.
"""
from __future__ import annotations

import numpy as np
import torch

from . import random as prng
from .Algorithm1 import _t
from .Algorithm3 import Algorithm3


class Algorithm2:
    def __init__(self, N_samples, N_iterations, observations, inputs, SSM, init_state_mean, init_state_cov, init_int_var_mean,
                 init_int_var_cov, GP_prior, basis_fcn, device=None):
        self.N_iterations = int(N_iterations)
        self.N_steps = np.asarray(observations).shape[0]
        self.cSMC = Algorithm3(N_samples=N_samples, observations=observations, inputs=inputs, SSM=SSM, init_state_mean=init_state_mean,
                               init_state_cov=init_state_cov, init_int_var_mean=init_int_var_mean, init_int_var_cov=init_int_var_cov,
                               GP_prior=GP_prior, basis_fcn=basis_fcn, device=device)   # :28-39

    def _trajectory_stats(self, state_traj, int_var_traj):
        """Statistics of one trajectory summed over time (:81-93, :146-160): (T0 (M,), T1 (M,M), T2, T3) per interface variable."""
        c = self.cSMC
        out = []
        for i in range(c.N_int):
            bf = c.basis_fcn[i]
            if hasattr(bf, "trajectory"):      # descriptors evaluate the whole trajectory in one batched call, row t with inputs[t]
                basis = bf.trajectory(state_traj, c.inputs.reshape(self.N_steps, -1))
            else:                              # arbitrary user callables keep the reference's per-step call
                basis = torch.stack([bf(state_traj[t:t + 1], c.inputs[t]).reshape(-1) for t in range(self.N_steps)])   # (T,M)
            if c.nvar[i] > 1:   # T0 (M, n), T2 (n, n)
                xi = int_var_traj[i].reshape(self.N_steps, c.nvar[i])
                out.append((basis.T @ xi, basis.T @ basis, xi.T @ xi, torch.tensor(float(self.N_steps), dtype=torch.float64, device=c.device)))
                continue
            xi = int_var_traj[i].reshape(-1)
            out.append((basis.T @ xi, basis.T @ basis, (xi * xi).sum(), torch.tensor(float(self.N_steps), dtype=torch.float64, device=c.device)))
        return out

    def __call__(self, key, init_ref_state, init_ref_int_var, progress=None):
        c, K, T, dev = self.cSMC, self.N_iterations, self.N_steps, self.cSMC.device
        nx = c.init_state_mean.numel()
        state_trace = torch.zeros((K, T, nx), dtype=torch.float64, device=dev)                                    # :46-54
        state_trace[0] = _t(init_ref_state.cpu() if isinstance(init_ref_state, torch.Tensor) else init_ref_state, dev).reshape(T, nx)
        int_var_trace = [torch.zeros((K, T, nv), dtype=torch.float64, device=dev) for nv in c.nvar]               # :56-67
        for i in range(c.N_int):
            v = init_ref_int_var[i]
            int_var_trace[i][0] = _t(v.cpu() if isinstance(v, torch.Tensor) else v, dev).reshape(T, c.nvar[i])
        sst = [[torch.zeros((K, M, nv), dtype=torch.float64, device=dev), torch.zeros((K, M, M), dtype=torch.float64, device=dev),
                torch.zeros((K, nv, nv), dtype=torch.float64, device=dev), torch.zeros((K,), dtype=torch.float64, device=dev)]
               for M, nv in zip(c.dim_basis, c.nvar)]   # :68-79
        ref_stats = self._trajectory_stats(state_trace[0], [v[0] for v in int_var_trace])                        # :81-93
        for i in range(c.N_int):
            for j in range(4):
                sst[i][j][0] = ref_stats[i][j].reshape(sst[i][j][0].shape)                                        # :94-99
        provider = hasattr(key, "student_t")
        key = key if provider else prng.as_key(key)
        for k in range(1, K):                                                                                     # :117-160
            if provider:
                key_step = key.fork(k)
            else:
                key, key_step = prng.split(key, 2)                                                                # :121
            new_state, new_int_var = c(key_step, state_trace[k - 1], [v[k - 1] for v in int_var_trace],
                                       [[sst[i][j][k - 1] for j in range(4)] for i in range(c.N_int)])            # :122-134
            state_trace[k] = new_state.reshape(T, nx)                                                             # :137
            stats = self._trajectory_stats(state_trace[k], [v.reshape(T, nv) for v, nv in zip(new_int_var, c.nvar)])
            for i in range(c.N_int):
                int_var_trace[i][k] = new_int_var[i].reshape(T, c.nvar[i])                                        # :139
                for j in range(4):
                    sst[i][j][k] = stats[i][j].reshape(sst[i][j][k].shape)                                        # :140-160
            if progress is not None:
                progress(k)
        state_trace = state_trace.transpose(0, 1).contiguous()                                                    # :161 -> (T,K,nx)
        int_var_trace = [v.transpose(0, 1).contiguous() for v in int_var_trace]                                   # :162-165
        obs_trace = torch.stack([c.SSM.output_mdl(state_trace[t], c.inputs[t], *[v[t] for v in int_var_trace]).reshape(K, -1) for t in range(T)])    # :168-173
        loglik = torch.stack([c.SSM.log_likelihood(c.observations[t], state_trace[t], c.inputs[t], *[v[t] for v in int_var_trace]) for t in range(T)])  # :176-185
        weights = torch.full((T, K), 1.0 / K, dtype=torch.float64, device=dev)                                    # :180
        return state_trace, int_var_trace, weights, sst, obs_trace, loglik
