"""Development probe: effective sample size and per-workgroup source-segment span along a sweep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments
T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N = 1 << 20
pb = experiments.smo_pgas(T=T)
pg = pgas_amd.PGAS(N, 2, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
ref = torch.as_tensor(pb.X_true, device="cuda")
A, S = pg.sample_params(pgas_amd.random.key(12345678), ref)
print("S", S.cpu().numpy().tolist(), "|A|max", float(A.abs().max()))
pg.cSMC(12345678, ref, A, S)
X, ANC, LW, _ = pg.cSMC.engine.traces()
w = torch.softmax(LW, 0)
print("final ESS", float(1 / (w * w).sum()), "of", N)
for t in (0, 1, 5, 20, 100, T // 2, T - 2):
    a = ANC[t].to(torch.int64)
    span = (a[1023::1024][:1023] - a[0::1024][:1023]).cpu().numpy() / 1024.0
    uniq = int(torch.unique(a).numel())
    print(f"t={t:5d} unique ancestors {uniq:8d}  segments spanned per workgroup: mean {span.mean():.2f} p99 {np.percentile(span,99):.1f} max {span.max():.1f}")
xs = X[-1]
print("state spread at T-1: std", xs.std(0).cpu().numpy(), "truth", pb.X_true[-1])
