"""Algorithm1 (marginalised online filter) at the reference's driver size: eager loop against the graph-replayed loop (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pgas_amd
from pgas_amd import experiments
N, T = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 750
pb = experiments.smo_marginal(T=T)
ssm = pb.ssm(pgas_amd.StateSpaceModel, torch)
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
