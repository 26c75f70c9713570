import sys, ctypes as C, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from common import canon_model, experiments, pgas_amd
from pgas_amd import _lib
N, T, ts = 70000, 40, 34
pb = experiments.smo_pgas(T=T)
A, S = experiments.initial_params(pb)
cm = canon_model(pb, N)
csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn)
LS, LSinv, cS = cm.chol_parts(S)
x = cm.init_state(12345678, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov), pb.X_true[0]); lw = None
for s in range(1, ts + 1):
    lwn, xn, an = cm.step(s, 12345678, x, lw, A, LS, LSinv, cS, pb.X_true[s])
    if s == ts:
        lwg, xg, ag = csmc.step(12345678, s, lw, x, A, S, pb.X_true[s])
        bad = np.nonzero(ag.cpu().numpy() != an)[0]
        print("bad", len(bad), bad[:6])
    x, lw = xn, lwn
out = np.zeros((4, 64))
L = _lib.load()
L.pgas_debug_dump(out.ctypes.data_as(C.POINTER(C.c_double)))
np.set_printoptions(linewidth=200, precision=17)
for w in range(4):
    print("wave", w, "S,b_lo,b_hi,ns,tau_f,tau_l,win_b0,nwin", out[w][:8])
print("cm[0:24]", out[0][8:32])
print("cand_b", out[0][32:40]); print("cand_cy", out[0][40:48]); print("cand_e", out[0][48:56]); print("cand_mp", out[0][56:64])
