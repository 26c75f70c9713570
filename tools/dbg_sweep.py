"""Development aid: run one sweep on the GPU and report where it first departs from the canonical oracle."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from common import canon_model, experiments, pgas_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
opts = {int(k): int(v) for k, v in (a.split("=") for a in sys.argv[3:])}
pb = experiments.smo_pgas(T=T)
A, S = experiments.initial_params(pb)
cm = canon_model(pb, N)
csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn, keep_logw_trace=True)
for k, v in opts.items():
    csmc.engine.set_option(k, v)
LS, LSinv, cS = cm.chol_parts(S)
csmc(12345678, pb.X_true, A, S)
X, ANC, LW, LT = csmc.engine.traces()
trajo, Xo, ANCo, lwo = cm.sweep(12345678, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov))
Xn, An = X.cpu().numpy(), ANC.cpu().numpy()
for t in range(T):
    bx = np.nonzero((Xn[t] != Xo[t]).any(axis=1))[0]
    ba = np.nonzero(An[t] != ANCo[t])[0] if t < T - 1 else []
    if len(bx) or len(ba):
        print(f"t={t}: {len(bx)} states differ (first {bx[:5]}), {len(ba)} ancestors differ (first {ba[:8]}, gpu {An[t][ba[:8]] if len(ba) else ''} oracle {ANCo[t][ba[:8]] if len(ba) else ''})")
        if len(ba):
            segs = np.unique(ba // 1024)
            print("   workgroups with wrong ancestors:", segs[:20], "count", len(segs))
        break
else:
    print("all equal", csmc.engine.launch_info())
# step API at the first failing step, teacher-forced with the oracle's own inputs
if 't' in dir() and t < T - 1 and (len(bx) or len(ba)):
    ts = t + 1   # ancestors row t belong to step t+1
    lwp = None
    x = Xo[0]
    lw = None
    for s in range(1, ts + 1):
        lwn, xn, an = cm.step(s, 12345678, x, lw, A, LS, LSinv, cS, pb.X_true[s])
        if s == ts:
            lwg, xg, ag = csmc.step(12345678, s, lw, x, A, S, pb.X_true[s])
            bad = np.nonzero(ag.cpu().numpy() != an)[0]
            print(f"step API at step {s}: {len(bad)} ancestors differ", bad[:8], ag.cpu().numpy()[bad[:8]], an[bad[:8]])
        x, lw = xn, lwn
