#!/usr/bin/env python3
"""bench.py -- particle-steps/sec of the conditional-SMC sweep on the SingleMassOscillator-PGAS
problem (BASELINE.json configs[1]: N = 2^20 particles, T = 2000, fp64, 1 x MI355X).

A "step" of this bench is ONE conditional-SMC sweep (reference condSequentialMonteCarlo.__call__,
src/PGAS.py:176-228): init, T-1 particle-filter steps with ancestor sampling, final index draw and
back-trace, all on the device, including every trace write.  value = N (T-1) sweeps / wall.

    python bench.py [--gpus G] [--steps K] [--warmup W] [--particles N] [--T T]

With G > 1 (launched by torch.distributed.run, one rank per GPU) every rank runs the sweep on its
own shard of work: see --mode.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
ALG_BYTES_PER_PARTICLE_STEP = 52  # 16 nx + 20, nx = 2


def cpu_baseline(pb, A, S, N, seed, steps):
    """Canonical C oracle (a port of the reference path, single thread) timed on this host for `steps` steps."""
    from oracle import canon

    bm, lik = pb.basis_fcn, pb.likelihood_fcn
    y = np.asarray(pb.observations, dtype=np.float64).reshape(pb.T, -1)
    u = np.asarray(pb.inputs, dtype=np.float64).reshape(pb.T, -1)
    cm = canon.CanonModel(N, pb.T, pb.nx, y.shape[1], u.shape[1], bm.basis.indices, bm.sel, bm.alpha, bm.beta, bm.basis.norm,
                          lik.H, lik.LRinv, lik.cR, y, u)
    LS, LSinv, cS = cm.chol_parts(S)
    L0 = np.linalg.cholesky(pb.init_state_cov)
    t0 = time.perf_counter()
    cm.sweep(seed, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, L0, traces=False, nsteps_limit=steps)
    dt = time.perf_counter() - t0
    return {
        "value": N * steps / dt, "unit": "particle-steps/s", "cores": 1, "kind": "port",
        "sample": f"canonical C oracle (oracle/pgas_canon.c, gcc -O2, 1 thread), N={N}, first {steps} of {pb.T - 1} steps, {dt:.1f} s; "
                  f"host has {os.cpu_count()} logical CPUs",
    }


def hbm_copy_rate(torch, device, nbytes=1 << 30, reps=10):
    """Attainable HBM rate of a plain device copy (read + write of `nbytes`), the yardstick SURVEY 8(d) asks for next to the 8 TB/s spec."""
    src = torch.empty(nbytes // 8, dtype=torch.float64, device=device).normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def bench_alg1(args, torch, dist, rank, local_rank, world):
    """Non-default workload: steps of the marginalised online filter (reference src/Algorithm1.py:297-397) at N particles per GPU;
    every rank filters its own particle set (independent replicas).  One bench step = one filter time step of all N particles."""
    import pgas_amd
    from pgas_amd import experiments

    N, K, W = args.particles, args.steps, args.warmup
    pb = experiments.smo_marginal(T=K + W + 6)
    ssm = pb.ssm(pgas_amd.StateSpaceModel, torch)
    alg = pgas_amd.Algorithm1(N_samples=N, observations=pb.observations, inputs=pb.inputs, SSM=ssm, forgetting_factor=pb.forgetting_factor,
                              init_state_mean=pb.init_state_mean, init_state_cov=pb.init_state_cov, init_int_var_mean=pb.init_int_var_mean,
                              init_int_var_cov=pb.init_int_var_cov, GP_prior=pb.GP_prior, basis_fcn=pb.basis_fcn(), device=f"cuda:{local_rank}")
    rand = alg._rand(12345678 + rank)
    st, iv, _, lw, _, ss = alg._init_algorithm(rand)
    x, l, v = st[0], lw[0], [iv[0][0]]
    t = 1
    for _ in range(W + 2):
        l, x, v, ss, a = alg.step(rand, t, l, x, v, ss)
        t += 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        l, x, v, ss, a = alg.step(rand, t, l, x, v, ss)
        t += 1
    barrier()
    dt = time.perf_counter() - t0
    alg.ops.check()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=alg.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    M = alg.dim_basis[0]

    def ev(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    out = {
        "metric": "particle-steps/sec, SingleMassOscillator marginalised online filter (Algorithm1) step", "value": N * K * world / dt,
        "unit": "particle-steps/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"SingleMassOscillator Algorithm1 step, N={N} particles/GPU, M={M} statistics per particle, fp64 "
                               f"(BASELINE.json configs[0]'s algorithm at scale)", "particles_per_gpu": N,
                   "parallelism": "1 GPU" if world == 1 else f"{world} independent particle sets, one per GPU"},
    }
    if rank == 0:
        P0, P1, _, _ = alg.GP_prior[0]
        phi = alg.basis_fcn[0](x, alg.inputs[t]).contiguous()
        xi = v[0].reshape(-1).contiguous()
        us_upd = ev(lambda: alg.ops.stats_gather_update(0.999, a, ss[0], phi, xi))
        us_fac = ev(lambda: alg.ops.mniw_solve(P0, P1, ss[0][0], ss[0][1], scale=0.999, phi=phi, want=("m", "q", "logdet"), keep_factor=True))
        b_upd = 2.0 * 8 * (M * M + M + 2) * N
        b_fac = 8.0 * (M * (M + 1) / 2 + (M + 2) * (M + 3) / 2 + 3 * M) * N
        out["roofline"] = {
            "bound": "hbm", "kernel": "k_mniw_solve_mfma", "achieved": b_fac / us_fac / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": b_fac / us_fac / 1e3 / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": us_fac,
            "note": "kernel durations of this non-default workload are measured with events around isolated launches after the timed region",
            "second_kernel": {"kernel": "k_stats_gather_update", "avg_launch_us": us_upd, "achieved_GBs": b_upd / us_upd / 1e3,
                              "frac": b_upd / us_upd / 1e3 / HBM_PEAK_GBS},
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--particles", type=int, default=1 << 20, help="particles per GPU")
    ap.add_argument("--T", type=int, default=2000)
    ap.add_argument("--cpu-steps", type=int, default=20, help="steps of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket the dominant kernel with HIP events")
    ap.add_argument("--chunk", type=int, default=-1, help="time steps per k_propagate launch (engine default if < 0)")
    ap.add_argument("--force-groups", action="store_true", help="run k_groups between the steps even where k_step could scan the groups itself")
    ap.add_argument("--no-overlap", action="store_true", help="run the weight recursion on the caller's stream (no concurrency)")
    ap.add_argument("--prop-lds", type=int, default=-1, help="LDS bytes reserved per k_propagate workgroup while overlapping (engine default if < 0)")
    ap.add_argument("--workload", choices=["smo", "vehicle", "emps", "smo-alg1"], default="smo",
                    help="smo = BASELINE configs[1] (the metric's configuration, default); vehicle = configs[2]; emps = configs[4]'s per-GPU chain; "
                         "smo-alg1 = the marginalised online filter (Algorithm1) on the SingleMassOscillator model, one filter step per bench step")
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas",
                    help="multi-GPU partition: independent chains, one per GPU (default; BASELINE config 5) or ONE sweep whose "
                         "--particles x G particles are sharded over the G ranks (RCCL all-gather per step + xGMI peer reads; config 4)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import pgas_amd
    from pgas_amd import experiments

    if args.workload == "smo-alg1":
        return bench_alg1(args, torch, dist, rank, local_rank, world)
    N, T = args.particles, args.T
    sharded_mode = args.mode == "sharded" and world > 1
    seed = 12345678 + (0 if sharded_mode else rank)  # independent chains differ by seed (BASELINE config 5 convention: 12345678 + g)
    pb = {"smo": experiments.smo_pgas, "vehicle": experiments.vehicle_pgas, "emps": experiments.emps_pgas}[args.workload](T=T)
    wl_name = {"smo": "SingleMassOscillator PGAS sweep, nx=2, M=41 Hilbert basis (BASELINE.json configs[1])",
               "vehicle": "Vehicle lateral dynamics PGAS sweep, nx=2, ny=2, M=729 3-D Hilbert basis (BASELINE.json configs[2], build's instantiation)",
               "emps": "EMPS PGAS sweep on synthetic data, nx=2, M=729 3-D Hilbert basis (one chain of BASELINE.json configs[4])"}[args.workload]
    pg = pgas_amd.PGAS(N, 2, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior,
                       pb.basis_fcn, device=f"cuda:{local_rank}")
    eng = pg.cSMC.engine
    ref = torch.as_tensor(pb.X_true, device=eng.device)
    # (A, S) from one sample_params on the initial reference trajectory
    A, S = pg.sample_params(pgas_amd.random.key(seed), ref)
    stride = int(os.environ.get("PGAS_PROF_STRIDE", "1"))       # time every stride-th launch of the profiled sweep(s)
    prof_all = os.environ.get("PGAS_PROF_ALL", "0") == "1"      # profile every timed sweep instead of the last one only
    if args.chunk >= 0:
        eng.set_option(1, args.chunk)      # PGAS_OPT_PROPAGATE_CHUNK
    if args.force_groups:
        eng.set_option(2, 1)               # PGAS_OPT_FORCE_SLOW_RESAMPLE
    if args.no_overlap:
        eng.set_option(3, 0)               # PGAS_OPT_OVERLAP
    if args.prop_lds >= 0:
        eng.set_option(4, args.prop_lds)   # PGAS_OPT_PROPAGATE_LDS

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    grp = None
    if sharded_mode:
        from pgas_amd import sharded

        # every rank must use the same (A, S): take rank 0's
        dist.broadcast(A, src=0)
        dist.broadcast(S, src=0)
        grp = sharded.make_dist_group(N * world, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn,
                                      pb.basis_fcn, device=f"cuda:{local_rank}")
        args.no_profile = True

    def one_sweep(sd):
        if grp is not None:
            sharded.sharded_sweep(grp, sd, ref, A, S, propagate_chunk=64)
        else:
            pg.cSMC(sd, ref, A, S)

    for w in range(args.warmup):
        one_sweep(seed + 1000 + w)
    barrier()
    t0 = time.perf_counter()
    prof_n, prof_ms, prop_n, prop_ms = 0, 0.0, 0, 0.0
    for k in range(args.steps):
        # kernel durations: the LAST timed sweep launches every k_resample_fast / k_propagate with start/stop HIP events
        # (hipExtLaunchKernelGGL: the dispatch's own begin/end timestamps, on the streams the kernels run on).  Timing every
        # launch of every sweep would cost ~8 % of the headline value; one sweep of K gives 2 x (T-1) samples inside the timed region.
        timed_sweep = not args.no_profile and (k == args.steps - 1 or prof_all)
        eng.set_profiling((-1 if stride == 1 else stride) if timed_sweep else 0)
        one_sweep(seed + k)
        if timed_sweep:
            n, ms, pn, pm = eng.profile()   # synchronises this sweep
            prof_n += n
            prof_ms += ms
            prop_n += pn
            prop_ms += pm
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=eng.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    units = N * (T - 1) * args.steps * world
    out = {
        "metric": "particle-steps/sec (N x (T-1) / wall), " + {"smo": "SingleMassOscillator", "vehicle": "Vehicle", "emps": "EMPS"}[args.workload] + " PGAS conditional-SMC sweep",
        "value": units / dt, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{wl_name}, N={N} particles/GPU, T={T}, fp64",
            "particles_per_gpu": N, "T": T,
            "parallelism": "1 GPU" if world == 1 else (
                f"one sweep of {N * world} particles sharded over {world} GPUs: RCCL all-gather of segment partials per step + xGMI peer reads"
                if sharded_mode else
                f"{world} independent chains, one per GPU, no data-path collective (replicas; --mode sharded runs the particle-sharded sweep)"),
        },
    }
    if rank == 0:
        if prof_n:
            # Dominant kernel (largest total duration in profiles/r01_kernel_stats.csv): k_resample_fast, one launch per time step,
            # N particle-steps per launch -> algorithmic bytes per launch = 52 N.  k_propagate (the other half of every
            # particle-step: basis, transition, noise) is reported next to it; it covers `chunk` time steps per launch.
            us = 1e3 * prof_ms / prof_n
            achieved = ALG_BYTES_PER_PARTICLE_STEP * N / (us * 1e-6) / 1e9
            p_us = 1e3 * prop_ms / max(prop_n, 1)
            p_steps = args.chunk if args.chunk > 0 else (1 if args.workload == "smo" else 16)   # time steps per k_propagate launch (engine default: 1 for 2-D bases, 16 for 3-D)
            traffic, p_traffic = None, None
            tf = os.path.join(ROOT, "profiles", "traffic_r01.json")
            if os.path.exists(tf):
                tj = json.load(open(tf))
                traffic, p_traffic = tj.get("k_resample_fast_hbm_bytes_per_launch"), tj.get("k_propagate_hbm_bytes_per_step")
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "kernel": "k_resample_fast", "avg_launch_us": us, "launches_timed": prof_n,
                "launch_sampling": f"every {'launch' if stride == 1 else str(stride) + 'th launch'} of {'every timed sweep' if prof_all else 'the last timed sweep'} "
                                   "carries start/stop HIP events (hipExtLaunchKernelGGL)",
                "alg_bytes_per_launch": ALG_BYTES_PER_PARTICLE_STEP * N,
                "hbm_copy_GBs": hbm_copy_rate(torch, eng.device),   # measured attainable rate of a 1 GiB device copy, outside the timed region
                "second_kernel": {"kernel": "k_propagate<2,2,8,2,2>" if args.workload == "smo" else "k_propagate<2,3,12,2,2>", "avg_launch_us": p_us, "launches_timed": prop_n, "steps_per_launch": p_steps,
                                  "achieved_GBs": ALG_BYTES_PER_PARTICLE_STEP * N * p_steps / (p_us * 1e-6) / 1e9,
                                  "traffic_per_step": p_traffic},
                "note": "the two kernels run concurrently on two streams; both are fp64-VALU-bound (DESIGN.md section 5), "
                        "so the HBM fraction understates how close they are to their own limit",
            }
        if args.cpu_steps > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(pb, A.cpu().numpy(), S.cpu().numpy(), N, seed, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
