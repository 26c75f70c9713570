import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments
N = 1 << 20
pb = experiments.emps_pgas(T=40)
A, S = experiments.initial_params(pb)
print("S", S.tolist(), "|A|", np.abs(A).max())
csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn)
csmc(1, pb.X_true, A, S); torch.cuda.synchronize()
X, ANC, LW, _ = csmc.engine.traces()
for t in (0, 1, 2, 5, 10, 20, 38):
    a = ANC[t].to(torch.int64)
    u = torch.unique(a)
    seg = torch.unique(a // 1024)
    span = (a[1023::1024][:1023] - a[0::1024][:1023]).cpu().numpy() / 1024.0
    print(f"t={t:3d} unique ancestors {u.numel():8d} in {seg.numel():5d} segments; per-workgroup span mean {span.mean():8.2f} max {span.max():8.1f}")
w = torch.softmax(LW, 0); print("ESS", float(1/(w*w).sum()))
