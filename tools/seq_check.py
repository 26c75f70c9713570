"""Sweep-to-sweep timing of two engines created one after the other in one process (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments
N = 1 << 20
for name, mk in (("SMO", lambda: experiments.smo_pgas(T=200)), ("EMPS", lambda: experiments.emps_pgas(T=100)), ("EMPS again", lambda: experiments.emps_pgas(T=100))):
    t0 = time.perf_counter(); pb = mk(); A, S = experiments.initial_params(pb); t1 = time.perf_counter()
    csmc = pgas_amd.condSequentialMonteCarlo(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts = []
    for k in range(6):
        a = time.perf_counter(); csmc(k, pb.X_true, A, S); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
        ts.append((1e3 * (b - a), 1e3 * (c - a)))
    print(f"{name}: setup {t1-t0:.2f}s create {t2-t1:.2f}s sweeps (enqueue ms, total ms):", " ".join(f"({e:.1f},{t:.1f})" for e, t in ts), flush=True)
    csmc.engine.close(); del csmc; torch.cuda.empty_cache()
