"""Diagnostic: phase timing inside k_step from the -DPG_STAMPS build (build/variants/lib_STAMPS.so), development aid.
usage: stamps_probe.py [T] [local] [overlap]"""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["PGAS_HIP_LIB"] = os.path.join(root, "build/variants/lib_STAMPS.so")
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments, _lib
T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N = 1 << 20
pb = experiments.smo_pgas(T=T)
pg = pgas_amd.PGAS(N, 2, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
ref = torch.as_tensor(pb.X_true, device="cuda")
A, S = pg.sample_params(pgas_amd.random.key(12345678), ref)
eng = pg.cSMC.engine
eng.set_option(3, 1 if "overlap" in sys.argv else 0)
eng.set_option(7, 1 if "local" in sys.argv else 0)
pg.cSMC(1, ref, A, S); torch.cuda.synchronize()
pg.cSMC(2, ref, A, S); torch.cuda.synchronize()
print(eng.launch_info(), "overlap" if "overlap" in sys.argv else "chain alone")
L = _lib.load()
buf = (ctypes.c_ulonglong * (2048 * 16))()
assert L.pgas_debug_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 16)[1:1025, :].astype(np.int64)   # block 0 is the ancestor workgroup
# stamps 0..5 and 7 come from the last launch (search only), 6 from the launch before it (search + scan)
t0 = st[:, 0].min()
rel = (st - t0) / 100.0  # us
names = ["start", "records+top level", "candidates", "staged", "searched", "gathered"]
print("phase boundaries of the last launch relative to the earliest workgroup start (us): median / p95 / max over workgroups")
for i, n in enumerate(names):
    print(f"  {i} {n:18s} {np.median(rel[:, i]):8.2f} {np.percentile(rel[:, i], 95):8.2f} {rel[:, i].max():8.2f}")
d = np.diff(st[:, :6], axis=1) / 100.0
print("per-phase durations (us), median over workgroups:", dict(zip(names[1:], np.round(np.median(d, axis=0), 2))))
print(f"end of the last launch (search only): median {np.median(rel[:, 7]):.2f}, max {rel[:, 7].max():.2f} us")
print(f"scan phase of the launch before (stamps 6 -> 8): median {np.median((st[:, 8] - st[:, 6]) / 100.0):.2f} us, p95 {np.percentile((st[:, 8] - st[:, 6]) / 100.0, 95):.2f}")
if st[:, 9].max() > 0:
    print("window_head detail (us after start, median): top scan done (wave 0) %.2f, after barrier %.2f, tab loads issued %.2f, window filled %.2f" % tuple(np.median(rel[:, i] - rel[:, 0]) for i in (9, 10, 11, 1)))
tot = rel[:, 5] - rel[:, 0]
print("start -> gathered per workgroup (us): " + "  ".join(f"p{q}: {np.percentile(tot, q):.2f}" for q in (50, 75, 90, 95, 99, 99.9)) + f"  max {tot.max():.2f}")
rounds = (st[:, 3] - st[:, 2]) / 100.0
print("candidates -> last staging (us): " + "  ".join(f"p{q}: {np.percentile(rounds, q):.2f}" for q in (50, 75, 90, 95, 99, 99.9)) + f"  max {rounds.max():.2f}")
print("workgroups by staging time: <2us", int((rounds < 2).sum()), " 2-5us", int(((rounds >= 2) & (rounds < 5)).sum()), " 5-8us", int(((rounds >= 5) & (rounds < 8)).sum()), " >=8us", int((rounds >= 8).sum()))
worst = np.argsort(tot)[-5:]
print("slowest workgroups (segment: boundaries)", {int(w): np.round(rel[w, :6] - rel[w, 0], 2).tolist() for w in worst})
