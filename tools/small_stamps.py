"""Diagnostic (-DPG_STAMPS build): cycle stamps of k_sweep_small's phases at t = 100, 101 (development aid)."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["PGAS_HIP_LIB"] = os.path.join(root, "build/variants/lib_STAMPS.so")
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments, _lib
N, T = int(os.environ.get("N", 200)), 400
pb = experiments.smo_pgas(T=T)
pg = pgas_amd.PGAS(N, 2, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
ref = torch.as_tensor(pb.X_true, device="cuda")
A, S = pg.sample_params(pgas_amd.random.key(12345678), ref)
pg.cSMC(1, ref, A, S); torch.cuda.synchronize()
L = _lib.load()
buf = (ctypes.c_ulonglong * (2048 * 16))()
assert L.pgas_debug_stamps(buf) == 0
p = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)[:16].reshape(2, 8)
print("k_sweep_small t=100 (cycles): propagation wait %d, scan %d, ancestor count %d, search+update+barrier %d; step %d -> next step's propagation done +%d"
      % (p[0,1]-p[0,0], p[0,2]-p[0,1], p[0,3]-p[0,2], p[0,4]-p[0,3], p[0,4]-p[0,0], p[1,0]-p[0,0]))
