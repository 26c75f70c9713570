"""Sweep time of the other BASELINE configurations at N = 2^20 (shortened T), for DESIGN.md (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments
N = 1 << 20
cfgs = (("SMO M=41 (2-D)", lambda: experiments.smo_pgas(T=200)), ("EMPS M=729 (3-D)", lambda: experiments.emps_pgas(T=100)),
        ("Vehicle M=729 (3-D, ny=2)", lambda: experiments.vehicle_pgas(T=100)), ("Toy M=40 (1-D, nx=1)", lambda: experiments.toy(T=40)))
for name, mk in [c for c in cfgs if os.environ.get("ONLY", "") in c[0]]:
    pb = mk()
    A, S = experiments.initial_params(pb)
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn,
                                             resample_before_propagate=bool(os.environ.get('CORRECTED')))
    if os.environ.get("NO_OVERLAP"):
        csmc.engine.set_option(3, 0)
    if os.environ.get("CHUNK"):
        csmc.engine.set_option(1, int(os.environ["CHUNK"]))
    csmc(1, pb.X_true, A, S); torch.cuda.synchronize()
    csmc.engine.set_profiling(int(os.environ.get('PROFILE', '3')))   # every 3rd launch: a stride that is not a divisor of the propagate chunk
    dt = 1e9
    for k in range(3):   # best of 3: a fresh engine occasionally sees one ~50 ms driver hiccup in its first sweeps (tools/seq_check.py)
        t0 = time.perf_counter(); csmc(2 + k, pb.X_true, A, S); torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t0)
    n, ms, pn, pm = csmc.engine.profile()
    print(f"{name:24s} T={pb.T:4d}: {1e3*dt:8.2f} ms/sweep = {1e6*dt/(pb.T-1):7.2f} us/step, {N*(pb.T-1)/dt:.3e} particle-steps/s; "
          f"k_step {1e3*ms/max(n,1):6.2f} us/launch, k_propagate {1e3*pm/max(pn,1):7.2f} us/launch")
    csmc.engine.close()
    del csmc
    torch.cuda.empty_cache()
