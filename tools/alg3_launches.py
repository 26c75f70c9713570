"""Kernel launches per step of the CONDITIONAL marginalised filter's eager loop (Algorithm3; development aid): run under
   rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/alg3_launches.py [T]
and divide the `Calls` column by T - 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pgas_amd
from pgas_amd import experiments
N, T = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 201
pb = experiments.smo_marginal(T=T)
ssm = pb.ssm_symbolic(pgas_amd.SymbolicStateSpaceModel)
a2 = pgas_amd.Algorithm2(N_samples=N, N_iterations=2, observations=pb.observations, inputs=pb.inputs, SSM=ssm, init_state_mean=pb.init_state_mean,
                         init_state_cov=pb.init_state_cov, init_int_var_mean=pb.init_int_var_mean, init_int_var_cov=pb.init_int_var_cov,
                         GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn())
c3 = a2.cSMC
orig = type(c3).__call__
type(c3).__call__ = lambda self, *a, _o=orig, **k: _o(self, *a, use_graph=False, **k)
a2(12345678, pb.X_true, list(pb.int_var_true))
torch.cuda.synchronize()
