"""Stage-by-stage GPU-vs-canonical-oracle comparison (development aid; run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
from common import *

def cmp(name, a, b):
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    a = a.reshape(b.shape)
    eq = np.array_equal(a, b)
    nbad = int((a != b).sum())
    mx = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.size else 0.0
    print(f"  {name:28s} {'BITEQ' if eq else 'DIFF '} nbad={nbad}/{a.size} max|d|={mx:.3e}", flush=True)
    return eq

def run(pb, N, seed=12345678, steps=3):
    print(f"== {pb.name} N={N} T={pb.T}", flush=True)
    A, S = experiments.initial_params(pb)
    cm = canon_model(pb, N)
    LS, LSinv, cS = cm.chol_parts(S)
    L0 = np.linalg.cholesky(pb.init_state_cov)
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn)
    eng = csmc.engine
    rng = np.random.default_rng(1)
    xs = pb.X_true[rng.integers(0, pb.T, 300)] + 0.01 * rng.standard_normal((300, pb.nx))
    ok = cmp("basis_eval", eng.basis_eval(xs, 3), cm.basis_eval(xs, 3))
    x0 = cm.init_state(seed, pb.init_state_mean, L0, pb.X_true[0])
    ok &= cmp("init_state", eng.init_state(seed, pb.X_true[0]), x0)
    eng.set_params(A, S)
    lw, x = None, x0
    for t in range(1, steps + 1):
        lwo, xo, ao, dbg = cm.step(t, seed, x, lw, A, LS, LSinv, cS, pb.X_true[t], debug=True)
        if t == 1:
            ok &= cmp("aux_states", eng.aux_states(x, t), dbg["aux"])
        lwg, xg, ag = csmc.step(seed, t, lw, x, A, S, pb.X_true[t])
        ok &= cmp(f"step{t} x_new", xg, xo)
        ok &= cmp(f"step{t} anc", ag, ao)
        ok &= cmp(f"step{t} logw", lwg, lwo)
        lw, x = lwo, xo
    t0 = time.time()
    traj = csmc(seed, pb.X_true, A, S)
    torch.cuda.synchronize()
    t1 = time.time()
    trajo, Xo, ANCo, lwlo = cm.sweep(seed, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, L0)
    t2 = time.time()
    X, ANC, LWL, _ = eng.traces()
    ok &= cmp("sweep x_trace", X, Xo)
    ok &= cmp("sweep anc_trace", ANC[: pb.T - 1], ANCo)
    ok &= cmp("sweep logw_last", LWL, lwlo)
    ok &= cmp("sweep traj", traj, trajo)
    print(f"  final idx {eng.last_final_index()}  gpu sweep {t1-t0:.3f}s  oracle {t2-t1:.3f}s  ALL {'OK' if ok else 'FAIL'}", flush=True)
    return ok

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    ok = True
    ok &= run(experiments.smo_pgas(T=30), 200)
    ok &= run(experiments.smo_pgas(T=30), 5000)
    ok &= run(experiments.toy(T=40), 1500)
    ok &= run(experiments.emps_pgas(T=12), 2048)
    ok &= run(experiments.smo_pgas(T=12), 1 << 17)
    print("OVERALL", "OK" if ok else "FAIL")
    sys.exit(0 if ok else 1)
