"""End-to-end Algorithm1 run at a large particle count (development aid): wall time and the learned force error."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments
N, T = int(sys.argv[1]), int(sys.argv[2])
pb = experiments.smo_marginal(T=T)
ssm = pb.ssm(pgas_amd.StateSpaceModel, torch)
alg = pgas_amd.Algorithm1(N_samples=N, observations=pb.observations, inputs=pb.inputs, SSM=ssm, forgetting_factor=pb.forgetting_factor,
                          init_state_mean=pb.init_state_mean, init_state_cov=pb.init_state_cov, init_int_var_mean=pb.init_int_var_mean,
                          init_int_var_cov=pb.init_int_var_cov, GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn())
torch.cuda.synchronize(); t0 = time.perf_counter()
out = alg(12345678)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
X, F, sst, w = out[0], out[1][0], out[2][0], out[3]
xm = (X[:, :, 0] * w).sum(1).cpu().numpy()
Fm = (F[:, :, 0] * w).sum(1).cpu().numpy()
print(f"Algorithm1 N={N} T={T}: {dt:.2f} s = {N*(T-1)/dt:.3e} particle-steps/s; position RMSE {np.sqrt(np.mean((xm[20:]-pb.X_true[20:,0])**2)):.4f}, "
      f"F_sd RMSE {np.sqrt(np.mean((Fm[T//2:]-pb.int_var_true[0][T//2:])**2)):.3f} (RMS {np.sqrt(np.mean(pb.int_var_true[0][T//2:]**2)):.3f}); "
      f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
