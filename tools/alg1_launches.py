"""Kernel launches per step of the marginalised filter's eager loop (development aid): run under
   rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/alg1_launches.py [T]   (SYMBOLIC=1: traced model callables)
and divide the `Calls` column by T - 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pgas_amd
from pgas_amd import experiments
N, T = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 201
pb = experiments.smo_marginal(T=T)
ssm = pb.ssm_symbolic(pgas_amd.SymbolicStateSpaceModel) if os.environ.get("SYMBOLIC") == "1" else pb.ssm(pgas_amd.StateSpaceModel, torch)
alg = pgas_amd.Algorithm1(N_samples=N, observations=pb.observations, inputs=pb.inputs, SSM=ssm, forgetting_factor=pb.forgetting_factor,
                          init_state_mean=pb.init_state_mean, init_state_cov=pb.init_state_cov, init_int_var_mean=pb.init_int_var_mean,
                          init_int_var_cov=pb.init_int_var_cov, GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn())
rand = alg._rand(12345678)
st, iv, sst, lw, anc, stats = alg._init_algorithm(rand)
traces = (st, iv, sst, lw, anc)
for t in range(1, T):
    stats = alg._loop_body(rand, t, traces, stats)
torch.cuda.synchronize()
