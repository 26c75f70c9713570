"""Sweep at N > 2^20 on one device (the k_resample + k_upper path), development aid."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments
N, T = int(sys.argv[1]), int(sys.argv[2])
pb = experiments.smo_pgas(T=T)
A, S = experiments.initial_params(pb)
csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn)
tr = csmc(1, pb.X_true, A, S); torch.cuda.synchronize()
t0 = time.perf_counter(); tr = csmc(2, pb.X_true, A, S); torch.cuda.synchronize(); dt = time.perf_counter() - t0
X, ANC, LW, _ = csmc.engine.traces()
ok = bool(torch.isfinite(tr).all()) and bool((ANC[:T-1, :-2] <= ANC[:T-1, 1:-1]).all())   # ancestors sorted (last column = reference particle)
print(f"N={N} T={T}: {1e3*dt:.1f} ms/sweep = {1e6*dt/(T-1):.1f} us/step = {N*(T-1)/dt:.3e} particle-steps/s; finite and sorted {ok}; "
      f"position RMSE of the sampled trajectory {float(torch.sqrt(((tr[:,0].cpu()-torch.as_tensor(pb.X_true[:,0]))**2).mean())):.4f}; "
      f"memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB (torch) ")
