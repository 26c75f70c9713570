"""Shared body of the reference's *_Simulation.py drivers for the marginalised family: Algorithm1 (online), a second Algorithm1 run
that supplies the initial reference trajectory (index drawn from the cumulative sum of the FLATTENED (T,N) weights, the reference's
quirk Q13, e.g. SingleMassOscillator_Simulation.py:54), then Algorithm2 (offline Particle Gibbs)."""
from __future__ import annotations

import os
import sys
import time as _time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_online_offline(pb, particles, iterations, seed=12345678, device=None, log=print):
    import torch

    import pgas_amd
    from pgas_amd import random as prng

    ssm = pb.ssm_symbolic(pgas_amd.SymbolicStateSpaceModel)   # the model callables traced into one-launch programs (StateSpaceModel + torch callables works the same)
    common = dict(observations=pb.observations, inputs=pb.inputs, SSM=ssm, init_state_mean=pb.init_state_mean, init_state_cov=pb.init_state_cov,
                  init_int_var_mean=pb.init_int_var_mean, init_int_var_cov=pb.init_int_var_cov, GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn(),
                  device=device)
    alg1 = pgas_amd.Algorithm1(N_samples=particles, forgetting_factor=pb.forgetting_factor, **common)
    alg2 = pgas_amd.Algorithm2(N_samples=particles, N_iterations=iterations, **common)
    T = pb.T
    key = prng.key(seed)
    key, key_sim = prng.split(key, 2)
    t0 = _time.perf_counter()
    online = alg1(key_sim)
    torch.cuda.synchronize()
    t_on = _time.perf_counter() - t0
    log(f"online (Algorithm1): N={particles}, T={T}: {t_on:.2f} s = {particles * (T - 1) / t_on:.3e} particle-steps/s")
    key, key_sim, key_traj = prng.split(key, 3)
    r_X, r_iv, _, r_w, r_anc, _, _, _ = alg1(key_sim)
    u = float(prng.uniform(key_traj, 1)[0])
    idx = min(int(np.searchsorted(np.cumsum(r_w.cpu().numpy()), u)), particles - 1)
    ref_state = pgas_amd.reconstruct_trajectory(r_X, r_anc, idx)
    ref_iv = [pgas_amd.reconstruct_trajectory(v, r_anc, idx) for v in r_iv]
    t0 = _time.perf_counter()
    offline = alg2(key, ref_state, ref_iv)
    torch.cuda.synchronize()
    t_off = _time.perf_counter() - t0
    log(f"offline (Algorithm2): K={iterations}: {t_off:.2f} s = {particles * (T - 1) * max(iterations - 1, 1) / t_off:.3e} particle-steps/s")
    return online, offline, dict(seconds_online=t_on, seconds_offline=t_off)


def posterior_mean(prior, stats):
    """MNIW posterior mean (1, M) of prior + stats (BI:35-45) for statistics given with the reference's shapes."""
    import pgas_amd

    e0 = np.asarray(prior[0]).reshape(-1, 1) + np.asarray(stats[0]).reshape(-1, 1)
    e1 = np.asarray(prior[1]) + np.asarray(stats[1])
    e2 = np.reshape(prior[2], (1, 1)) + np.reshape(stats[2], (1, 1))
    e3 = float(np.reshape(prior[3], -1)[0]) + float(np.reshape(stats[3], -1)[0])
    return pgas_amd.prior_mniw_2naturalPara_inv(e0, e1, e2, e3)[0]
