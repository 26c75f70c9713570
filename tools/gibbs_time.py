"""Per-Gibbs-iteration time of PGAS at the BASELINE size (sweep + sample_params), for DESIGN.md (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments
N, T, K = 1 << 20, 2000, 4
pb = experiments.smo_pgas(T=T)
pg = pgas_amd.PGAS(N, K, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
ref = torch.as_tensor(pb.X_true, device="cuda")
t0 = time.perf_counter(); A, S = pg.sample_params(pgas_amd.random.key(1), ref); torch.cuda.synchronize(); print("sample_params (first, incl. allocations) %.2f ms" % (1e3 * (time.perf_counter() - t0)))
t0 = time.perf_counter(); A, S = pg.sample_params(pgas_amd.random.key(2), ref); torch.cuda.synchronize(); print("sample_params %.2f ms" % (1e3 * (time.perf_counter() - t0)))
pg(pgas_amd.random.key(3), pb.X_true)   # warm-up run of the whole loop (K-1 sweeps)
torch.cuda.synchronize()
t0 = time.perf_counter(); trace, ll = pg(pgas_amd.random.key(12345678), pb.X_true); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("PGAS.__call__ K=%d: %.1f ms total = %.1f ms per Gibbs iteration (sweep + sample_params); trace %s" % (K, 1e3 * dt, 1e3 * dt / (K - 1), tuple(trace.shape)))
