"""Per-step time of the marginalised online filter (Algorithm1) and of its HIP kernels at several particle counts (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pgas_amd
from pgas_amd import experiments

def ev(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us

MODEL = os.environ.get("MODEL", "smo")
MB = int(os.environ.get("M", "0"))
for N in [int(a) for a in (sys.argv[1:] or ["200", "16384", "131072", "1048576"])]:
    T = 24 if N > 100000 else 60
    pb = experiments.smo_marginal(T=T) if MODEL == "smo" else experiments.vehicle_marginal(T=T, M=MB or 20)
    ssm = pb.ssm(pgas_amd.StateSpaceModel, torch)
    alg = pgas_amd.Algorithm1(N_samples=N, observations=pb.observations, inputs=pb.inputs, SSM=ssm, forgetting_factor=pb.forgetting_factor,
                              init_state_mean=pb.init_state_mean, init_state_cov=pb.init_state_cov, init_int_var_mean=pb.init_int_var_mean,
                              init_int_var_cov=pb.init_int_var_cov, GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn())
    alg.ops.eng.set_option(6, int(os.environ.get('MNIW_VALU', '0')))
    rand = alg._rand(1)
    st, iv, sst, lw, anc, ss = alg._init_algorithm(rand)
    x, l, v = st[0], lw[0], [iv[i][0] for i in range(len(iv))]
    for t in range(1, 4):
        l, x, v, ss, a = alg.step(rand, t, l, x, v, ss)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nst = T - 5
    for t in range(4, 4 + nst):
        l, x, v, ss, a = alg.step(rand, t, l, x, v, ss)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nst
    M = alg.dim_basis[0]
    ops, (P0, P1, _, _) = alg.ops, alg.GP_prior[0]
    phi = alg.basis_fcn[0](x, alg.inputs[5]).contiguous()
    xi = v[0].reshape(-1).contiguous()
    us_solve = ev(lambda: ops.mniw_solve(P0, P1, ss[0][0], ss[0][1], scale=0.999, anc=a, phi=phi, want=("m", "c", "q")))
    fac = ops.mniw_solve(P0, P1, ss[0][0], ss[0][1], scale=0.999, phi=phi, want=("m", "q", "logdet"), keep_factor=True)
    us_fac = ev(lambda: ops.mniw_solve(P0, P1, ss[0][0], ss[0][1], scale=0.999, phi=phi, want=("m", "q", "logdet"), keep_factor=True))
    us_tri = ev(lambda: ops.mniw_trisolve(fac, a, phi))
    us_upd = ev(lambda: ops.stats_gather_update(0.999, a, ss[0], phi, xi))
    us_rs = ev(lambda: ops.systematic_resample(0.3, l))
    w = torch.softmax(l, 0)
    us_wsum = ev(lambda: alg._weighted(ss[0], w))
    bytes_stats = 8.0 * (M * M + M + 2) * N
    print(f"{pb.name} M={M} x{len(iv)} N={N:8d}: step {dt*1e3:8.3f} ms = {N/dt:.3e} particle-steps/s | k_mniw_solve+store {us_fac:9.1f} us | k_mniw_trisolve {us_tri:8.1f} us ({8.0*((M+2)*(M+3)/2+M)*N/us_tri/1e3:7.1f} GB/s) | k_mniw_solve {us_solve:9.1f} us ({bytes_stats/us_solve/1e3:7.1f} GB/s read) | "
          f"k_stats_gather_update {us_upd:9.1f} us ({2*bytes_stats/us_upd/1e3:7.1f} GB/s r+w) | resample {us_rs:7.1f} us | k_weighted_stats {us_wsum:9.1f} us ({bytes_stats/us_wsum/1e3:7.1f} GB/s read)", flush=True)
    del alg, ss, st, iv, sst, lw, anc
    torch.cuda.empty_cache()
