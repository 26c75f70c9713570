"""Algorithm1 (marginalised online filter) at the reference's driver size: eager loop against the graph-replayed loop (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pgas_amd
from pgas_amd import experiments
N, T = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 750
pb = experiments.smo_marginal(T=T)
# SYMBOLIC=1: the model's callables as traced one-launch programs (pgas_amd.SymbolicStateSpaceModel) instead of torch operations
ssm = pb.ssm_symbolic(pgas_amd.SymbolicStateSpaceModel) if os.environ.get("SYMBOLIC") == "1" else pb.ssm(pgas_amd.StateSpaceModel, torch)
print("model callables:", type(ssm).__name__, flush=True)
alg = pgas_amd.Algorithm1(N_samples=N, observations=pb.observations, inputs=pb.inputs, SSM=ssm, forgetting_factor=pb.forgetting_factor,
                          init_state_mean=pb.init_state_mean, init_state_cov=pb.init_state_cov, init_int_var_mean=pb.init_int_var_mean,
                          init_int_var_cov=pb.init_int_var_cov, GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn())
for mode in (False, True, False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = alg(12345678, use_graph=mode)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"N={N} T={T} use_graph={mode}: {dt:.3f} s = {1e3 * dt / (T - 1):.3f} ms per step (whole __call__, incl. the obs / log-likelihood passes after the loop)", flush=True)
# the loop alone (what the graph replaces), without the trace post-processing of __call__
for mode in (False, True):
    rand = alg._rand(12345678)
    st, iv, sst, lw, anc, stats = alg._init_algorithm(rand)
    traces = (st, iv, sst, lw, anc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode:
        alg._graphed_loop(rand, traces, stats, T)
    else:
        for t in range(1, T):
            stats = alg._loop_body(rand, t, traces, stats)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"loop only, use_graph={mode}: {1e3 * dt / (T - 1):.3f} ms per step", flush=True)
# Algorithm2 (Particle Gibbs over the conditional filter): seconds per Gibbs iteration, graph replay against the eager loop
import numpy as np
K = 3
common = dict(N_samples=N, N_iterations=K, observations=pb.observations, inputs=pb.inputs, SSM=ssm, init_state_mean=pb.init_state_mean,
              init_state_cov=pb.init_state_cov, init_int_var_mean=pb.init_int_var_mean, init_int_var_cov=pb.init_int_var_cov,
              GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn())
a2 = pgas_amd.Algorithm2(**common)
for mode in (True, False):
    c3 = a2.cSMC
    orig = type(c3).__call__
    type(c3).__call__ = lambda self, *a, _o=orig, _m=mode, **k: _o(self, *a, use_graph=_m, **k)
    try:
        a2(1, pb.X_true, list(pb.int_var_true)); torch.cuda.synchronize()
        t0 = time.perf_counter(); a2(12345678, pb.X_true, list(pb.int_var_true)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    finally:
        type(c3).__call__ = orig
    print(f"Algorithm2 N={N} T={T} K={K} use_graph={mode}: {dt / (K - 1):.3f} s per Gibbs iteration", flush=True)
