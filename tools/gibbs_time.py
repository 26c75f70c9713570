"""Per-Gibbs-iteration time of PGAS at the BASELINE size (sweep + sample_params) and the pieces of sample_params, for DESIGN.md
(development aid).  usage: gibbs_time.py [smo|emps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments
which = sys.argv[1] if len(sys.argv) > 1 else "smo"
N, T, K = 1 << 20, 2000, 3
pb = experiments.smo_pgas(T=T) if which == "smo" else experiments.emps_pgas(T=T)
pg = pgas_amd.PGAS(N, K, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
eng = pg.cSMC.engine
ref = torch.as_tensor(pb.X_true, device="cuda")


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, r


ms, _ = timed(lambda: pg.sample_params(pgas_amd.random.key(2), ref))
print(f"{which}: M = {eng.M}; sample_params {ms:.3f} ms")
ms_s, (T0, T1, T2, T3) = timed(lambda: eng.suffstats(ref))
e0, e1 = pg.GP_prior[0] + T0, pg.GP_prior[1] + T1
ms_c, Lc = timed(lambda: torch.linalg.cholesky(e1))
eye = torch.eye(eng.M, dtype=torch.float64, device="cuda")
ms_v, sol = timed(lambda: torch.cholesky_solve(torch.cat([e0, eye], dim=1), Lc))
ms_c2, _ = timed(lambda: torch.linalg.cholesky(sol[:, eng.nx:]))
ms_d, _ = timed(lambda: pg.param_draws(pgas_amd.random.key(5)))
print(f"  pieces: suff-stats (3 HIP kernels) {ms_s:.3f} ms, cholesky(eta1) {ms_c:.3f} ms, cholesky_solve([eta0 | I]) {ms_v:.3f} ms, cholesky(col_cov) {ms_c2:.3f} ms, "
      f"device draws {ms_d:.3f} ms")
pg(pgas_amd.random.key(3), pb.X_true)   # warm-up run of the whole loop (K-1 sweeps)
torch.cuda.synchronize()
t0 = time.perf_counter(); trace, ll = pg(pgas_amd.random.key(12345678), pb.X_true); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"  PGAS.__call__ K={K}: {1e3 * dt:.1f} ms total = {1e3 * dt / (K - 1):.1f} ms per Gibbs iteration (sweep + sample_params): sample_params is {100 * ms / (1e3 * dt / (K - 1)):.1f} % of it")
