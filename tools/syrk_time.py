"""pgas_suffstats timing at full size (EMPS M = 729 and SMO M = 41, T = 2000): wall per call by HIP events, sweep over the row splits
(development aid; per-kernel times come from rocprofv3 --kernel-trace --stats of this script)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments

for name, mk in [c for c in (("EMPS M=729", lambda: experiments.emps_pgas(T=2000)), ("SMO M=41", lambda: experiments.smo_pgas(T=2000))) if os.environ.get("ONLY", "") in c[0]]:
    pb = mk()
    pg = pgas_amd.PGAS(1024, 2, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior, pb.basis_fcn)
    eng = pg.cSMC.engine
    traj = torch.as_tensor(pb.X_true, device="cuda")
    M = pg.cSMC.engine.M if hasattr(pg.cSMC.engine, "M") else None
    for S in [int(a) for a in sys.argv[1:]] or [0]:
        eng.set_option(10, S)
        for _ in range(3):
            out = eng.suffstats(traj)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            out = eng.suffstats(traj)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        Mv = out[1].shape[0]
        fl = (pb.T - 1) * Mv * (Mv + 1)   # multiply-adds x 2 of the lower triangle of Phi^T Phi
        print(f"{name}: splits {S or 'auto'}: {us:8.1f} us per pgas_suffstats call (3 kernels); triangle flops {fl:.3e} -> {fl / us / 1e6:.2f} TFLOP/s over the whole call", flush=True)
    eng.close()
