"""Diagnostic: phase timing inside k_resample_fast from the -DPG_STAMPS build (build/ablate/lib_STAMPS.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PGAS_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build/ablate/lib_STAMPS.so")
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments, _lib
T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N = 1 << 20
pb = experiments.smo_pgas(T=T)
pg = pgas_amd.PGAS(N, 2, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
ref = torch.as_tensor(pb.X_true, device="cuda")
A, S = pg.sample_params(pgas_amd.random.key(12345678), ref)
pg.cSMC.engine.set_option(3, 0)
pg.cSMC(1, ref, A, S); torch.cuda.synchronize()
pg.cSMC(2, ref, A, S); torch.cuda.synchronize()
L = _lib.load()
buf = (ctypes.c_ulonglong * (2048 * 16))()
assert L.pgas_debug_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 16)[1:1025, :].astype(np.int64)
# stamps are from the LAST launch that executed each phase (launch T has no scan: stamps 6,7 from launch T-1 / T)
t0 = st[:, 0].min()
rel = (st - t0) / 100.0  # us
names = ["start", "upper done", "range/cands", "staged", "searched", "gathered", "pre-scan", "end"]
print("phase boundaries relative to the earliest workgroup start (us): median / p95 / max over workgroups")
for i, n in enumerate(names):
    print(f"  {i} {n:12s} {np.median(rel[:, i]):8.2f} {np.percentile(rel[:, i], 95):8.2f} {rel[:, i].max():8.2f}")
order = [0, 8, 9, 10, 11, 12, 13, 1]
labels = ["loads+max", "barrier1", "exp+scan0", "barrier2", "lvl1+excl+max", "barrier3", "carry"]
du = np.diff(st[:, order], axis=1) / 100.0
print("inside upper_core (us, median):", dict(zip(labels, np.round(np.median(du, axis=0), 2))))
d = np.diff(st[:, :6], axis=1) / 100.0
print("per-phase durations (us), median over workgroups:", np.round(np.median(d, axis=0), 2))
print("last workgroup:", np.round((st[1023, :6] - t0) / 100.0, 2))
