"""How long does the host need to ENQUEUE one sweep (pgas_sweep returns without synchronising) against how long the device needs to run it?"""
import sys, time
import torch
sys.path.insert(0, ".")
import pgas_amd
from pgas_amd import experiments
import os
N, T = int(os.environ.get("N", 1 << 20)), int(os.environ.get("T", 2000))
pb = experiments.smo_pgas(T=T)
pg = pgas_amd.PGAS(N, 2, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
ref = torch.as_tensor(pb.X_true, device=pg.cSMC.engine.device)
A, S = pg.sample_params(pgas_amd.random.key(1), ref)
for opt in sys.argv[1:]:
    k, v = opt.split("=")
    pg.cSMC.engine.set_option(int(k), int(v))
pg.cSMC(1, ref, A, S)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    pg.cSMC(2 + rep, ref, A, S)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0):.1f} ms, until done {1e3 * (t2 - t0):.1f} ms")
