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
FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X_MICROARCH.md: fp64 vector (= fp64 matrix) peak


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
    ssm = pb.ssm_symbolic(pgas_amd.SymbolicStateSpaceModel)   # the model callables traced into one-launch programs (StateSpaceModel + torch callables works the same)
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


def numpy_baseline_config1(pb, A, S, seed):
    """BASELINE.md config 1 ("plumbing"): one full conditional-SMC sweep at the reference's own particle count N = 200, T = pb.T,
    timed with the literal NumPy restatement of the reference (oracle/pgas_numpy.py) and with the canonical C oracle."""
    from oracle import canon, pgas_numpy

    N, T, nx = 200, pb.T, pb.nx
    bm, lik = pb.basis_fcn, pb.likelihood_fcn
    b = bm.basis
    y = np.asarray(pb.observations, dtype=np.float64).reshape(T, -1)
    u = np.asarray(pb.inputs, dtype=np.float64).reshape(T, -1)
    eig = (np.pi * b.indices.astype(np.float64) / b.size) ** 2     # src/BasisFunctions.py:60
    L = b.size / 2

    def basis(state, ut):
        state = np.atleast_2d(state)
        v = state if not np.size(ut) else np.hstack([state, np.broadcast_to(np.atleast_1d(ut), (state.shape[0], np.size(ut)))])
        xc = v[:, bm.sel] / bm.div - b.center
        return np.prod(np.sqrt(1 / L) * np.sin(np.sqrt(eig)[None] * (xc[:, None, :] + L)), axis=2)  # :77-80

    def likelihood(obs, state, ut):
        return pgas_numpy.mvn_logpdf(np.atleast_1d(obs), np.atleast_2d(state) @ lik.H.T, lik.R)

    csmc = pgas_numpy.condSequentialMonteCarlo(N, y, u, pb.init_state_mean, pb.init_state_cov, likelihood, basis)
    z = np.zeros((T, N, nx))
    for t in range(1, T):
        z[t] = canon.normals(seed, canon.STREAM_PROP, t, 0, N, nx)
    rand = dict(z0=canon.normals(seed, canon.STREAM_INIT, 0, 0, N, nx), z=z,
                u_resample=np.array([canon.uniform(seed, canon.STREAM_RESAMPLE, t) for t in range(T)]),
                u_ancestor=np.array([canon.uniform(seed, canon.STREAM_ANCESTOR, t) for t in range(T)]),
                u_final=canon.uniform(seed, canon.STREAM_FINAL, 0))
    t0 = time.perf_counter()
    csmc(rand, pb.X_true, A, S)
    dt_np = time.perf_counter() - t0
    cm = canon.CanonModel(N, T, nx, y.shape[1], u.shape[1], b.indices, bm.sel, bm.alpha, bm.beta, b.norm, lik.H, lik.LRinv, lik.cR, y, u)
    LS, LSinv, cS = cm.chol_parts(S)
    t0 = time.perf_counter()
    cm.sweep(seed, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov))
    dt_c = time.perf_counter() - t0
    blas = os.environ.get("OMP_NUM_THREADS", "unset")
    return {
        "workload": f"BASELINE.json configs[0]: N={N}, T={T} full sweep incl. traces and back-trace, fp64",
        "numpy_restatement": {"value": N * (T - 1) / dt_np, "unit": "particle-steps/s", "seconds": dt_np, "cores": 1, "kind": "port",
                              "note": f"oracle/pgas_numpy.py (literal NumPy restatement of src/PGAS.py:176-228), single process, NumPy {np.__version__}, "
                                      f"OMP_NUM_THREADS={blas}, host has {os.cpu_count()} logical CPUs (arrays of 200 x 41: no BLAS threading to speak of)"},
        "c_oracle": {"value": N * (T - 1) / dt_c, "unit": "particle-steps/s", "seconds": dt_c, "cores": 1, "kind": "port",
                     "note": "oracle/pgas_canon.c, gcc -O2, 1 thread"},
    }


def respawn_under_torchrun(args):
    """`python bench.py --gpus G` with G > 1 and no launcher: start G ranks (one per GPU) before this process touches the GPU,
    stream their output and exit with their code.  The driver's own torch.distributed.run launch does not come through here."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: --gpus {args.gpus} without a launcher (WORLD_SIZE unset): starting {args.gpus} ranks with torch.distributed.run", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd))


def verify_last_sweep(torch, eng, ref, traj, owns_conditioned, chase):
    """What the timed region produced must be a conditional-SMC sweep, not just kernels that ran: the trajectory is finite, the final
    log-weights are finite, the conditioned particle followed the reference at every time (src/PGAS.py:194,214), and (chase=True:
    unsharded contexts, whose traces this process can index) the trajectory is a path through the state trace along the recorded
    ancestors (src/Filtering.py:40-55), chased here with plain torch indexing from the final index.  Returns a list of failures."""
    bad = []
    T, nx = traj.shape
    if not bool(torch.isfinite(traj).all()):
        bad.append("trajectory has non-finite entries")
    px = eng.traces_blocks(eng.TRACE_X, (eng.N, nx), torch.float64)
    pa = eng.traces_blocks(eng.TRACE_ANC, (eng.N,), torch.int32)
    if owns_conditioned:
        t0 = 0
        for blk in px:
            if not torch.equal(blk[:, -1, :], ref[t0:t0 + blk.shape[0]]):
                bad.append(f"conditioned particle left the reference trajectory in rows {t0}..{t0 + blk.shape[0] - 1}")
            t0 += blk.shape[0]
    _, _, lw, _ = eng.traces(copy_blocks=False)
    if not bool(torch.isfinite(lw).all()):
        bad.append("final log-weights are not finite")
    if chase:
        rx, ra = px[0].shape[0], pa[0].shape[0]
        b = torch.tensor([eng.last_final_index()], device=traj.device, dtype=torch.int64)
        ok = torch.ones((), dtype=torch.bool, device=traj.device)
        for t in range(T - 1, -1, -1):
            ok = ok & (px[t // rx][t % rx].index_select(0, b)[0] == traj[t]).all()
            if t:
                b = pa[(t - 1) // ra][(t - 1) % ra].index_select(0, b).to(torch.int64)
        if not bool(ok):
            bad.append("the trajectory is not the ancestral path of the final index through the traces")
    return bad


def sharded_self_check(torch, dist, pgas_amd, experiments, sharded, local_rank, rank, world):
    """Before anything is timed in sharded mode: a small sharded sweep (8192 particles per rank, T = 12: remote ancestors, the RCCL
    all-gather, IPC peer reads, the cross-rank ancestor chase) against the SAME sweep on an unsharded context on every rank; the
    trajectories must be bit-identical on every rank.  Returns None or the failure text (identical on every rank)."""
    Ns, Ts = 8192, 12
    pbs = experiments.smo_pgas(T=Ts)
    A, S = experiments.initial_params(pbs)
    dev = f"cuda:{local_rank}"
    err = None
    try:
        g = sharded.make_dist_group(Ns * world, pbs.observations, pbs.inputs, pbs.init_state_mean, pbs.init_state_cov, pbs.likelihood_fcn, pbs.basis_fcn, device=dev)
        tr = sharded.sharded_sweep(g, 4242, pbs.X_true, A, S).clone()
        one = pgas_amd.condSequentialMonteCarlo(Ns * world, pbs.observations, pbs.inputs, pbs.init_state_mean, pbs.init_state_cov, pbs.likelihood_fcn,
                                                pbs.basis_fcn, device=dev)
        tu = one(4242, pbs.X_true, A, S)
        torch.cuda.synchronize()
        if not torch.equal(tr, tu):
            err = f"rank {rank}: sharded trajectory differs from the unsharded one in {int((tr != tu).any(dim=1).sum())} of {Ts} rows"
        g.barrier()
        g.shards[0].eng.close()
        one.engine.close()
    except Exception as e:   # noqa: BLE001 -- agreed on below
        err = f"rank {rank}: {type(e).__name__}: {e}"
    errs = [None] * world
    dist.all_gather_object(errs, err)
    bad = [e for e in errs if e]
    return "; ".join(bad) if bad else None


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
    ap.add_argument("--tail-groups", action="store_true", help="group scans in k_step's tail (in-launch hand-off) instead of k_groups launches between the steps")
    ap.add_argument("--local-groups", action="store_true", help="k_step<LOCAL>: every workgroup scans all groups itself instead of a k_groups launch between the steps")
    ap.add_argument("--event-stride", type=int, default=-1, help="k_propagate launches per gating event (engine default if < 0)")
    ap.add_argument("--max-lead", type=int, default=-1, help="PGAS_OPT_MAX_LEAD: event groups k_propagate may run ahead of the weight recursion (engine default if < 0)")
    ap.add_argument("--no-overlap", action="store_true", help="run the weight recursion on the caller's stream (no concurrency)")
    ap.add_argument("--prop-lds", type=int, default=-1, help="LDS bytes reserved per k_propagate workgroup while overlapping (engine default if < 0)")
    ap.add_argument("--workload", choices=["smo", "vehicle", "emps", "smo-alg1"], default="smo",
                    help="smo = BASELINE configs[1] (the metric's configuration, default); vehicle = configs[2]; emps = configs[4]'s per-GPU chain; "
                         "smo-alg1 = the marginalised online filter (Algorithm1) on the SingleMassOscillator model, one filter step per bench step")
    ap.add_argument("--mode", choices=["auto", "replicas", "sharded"], default="auto",
                    help="multi-GPU partition.  sharded: ONE sweep whose --particles x G particles are sharded over the G ranks (RCCL all-gather of the "
                         "segment partials per step + xGMI peer reads; BASELINE config 4, the north-star split).  replicas: independent chains, one per GPU "
                         "(BASELINE config 5).  auto (default): sharded for the smo workload on G > 1 GPUs, replicas for the emps / vehicle / smo-alg1 workloads")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not os.environ.get("PGAS_BENCH_REHEARSE"):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or without a launcher)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    # PGAS_BENCH_REHEARSE=1 (development aid, one-GPU boxes): every rank uses device 0 and the collectives run over gloo with the
    # library's host-callback all-gather -- the multi-rank code path of this script (IPC peer mappings between processes included) with
    # everything but RCCL and xGMI.  The numbers of such a run mean nothing.
    rehearse = os.environ.get("PGAS_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("PGAS_BENCH_REHEARSE_DUMP_S", "90")), repeat=True, file=sys.stderr)   # where is a stalled rank?
    torch.cuda.set_device(local_rank)
    # a launcher (torch.distributed.run) with ONE rank and an explicit --mode sharded runs the sharded code path with world = 1:
    # the same kernels and one-rank RCCL collectives, no wire time -- the figure to hold against the unsharded sweep
    use_dist = world > 1 or ("WORLD_SIZE" in os.environ and args.mode == "sharded")
    if use_dist and not rehearse:
        # a multi-rank run that stops making progress (a collective nobody answers, a driver call that never returns) should end with a
        # stack trace and a non-zero exit, not sit there until someone's timeout: setup + 6 sweeps take well under a minute
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("PGAS_BENCH_WATCHDOG_S", "900")), exit=True, file=sys.stderr)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import pgas_amd
    from pgas_amd import experiments

    if args.workload == "smo-alg1":
        return bench_alg1(args, torch, dist, rank, local_rank, world)
    N, T = args.particles, args.T
    mode = args.mode if args.mode != "auto" else ("sharded" if args.workload == "smo" else "replicas")
    sharded_mode = mode == "sharded" and use_dist
    seed = 12345678 + (0 if sharded_mode else rank)  # independent chains differ by seed (BASELINE config 5 convention: 12345678 + g)
    pb = {"smo": experiments.smo_pgas, "vehicle": experiments.vehicle_pgas, "emps": experiments.emps_pgas}[args.workload](T=T)
    wl_name = {"smo": "SingleMassOscillator PGAS sweep, nx=2, M=41 Hilbert basis (BASELINE.json configs[1])",
               "vehicle": "Vehicle lateral dynamics PGAS sweep, nx=2, ny=2, M=729 3-D Hilbert basis (BASELINE.json configs[2], build's instantiation)",
               "emps": "EMPS PGAS sweep on synthetic data, nx=2, M=729 3-D Hilbert basis (one chain of BASELINE.json configs[4])"}[args.workload]
    if sharded_mode and args.workload == "smo":
        wl_name = "SingleMassOscillator PGAS sweep, nx=2, M=41 Hilbert basis, particles sharded over the GPUs (BASELINE.json configs[3] at 8 GPUs)"
    pg = pgas_amd.PGAS(N, 2, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.GP_prior,
                       pb.basis_fcn, device=f"cuda:{local_rank}")
    ref = torch.as_tensor(pb.X_true, device=pg.cSMC.engine.device)
    # (A, S) from one sample_params on the initial reference trajectory
    A, S = pg.sample_params(pgas_amd.random.key(seed), ref)
    stride = int(os.environ.get("PGAS_PROF_STRIDE", "4"))       # time every stride-th launch of the profiled sweep(s)
    prof_all = os.environ.get("PGAS_PROF_ALL", "0") == "1"      # profile every timed sweep instead of the last one only

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def mark(msg):
        if rehearse:
            print(f"[rehearse rank {rank}] {msg}", file=sys.stderr, flush=True)

    mark("model and parameters ready")
    grp = None
    self_check = None
    if sharded_mode:
        from pgas_amd import sharded

        # every rank must use the same (A, S): take rank 0's
        dist.broadcast(A, src=0)
        dist.broadcast(S, src=0)
        # A sharded run that cannot be set up, or whose small self-check disagrees with the unsharded sweep, ENDS here with a non-zero exit
        # on every rank: independent chains measured under the sharded metric would put a wrong point into a scaling table.
        # (`--mode replicas` measures independent chains on purpose.)
        if not os.environ.get("PGAS_BENCH_NO_SELF_CHECK"):
            failed = sharded_self_check(torch, dist, pgas_amd, experiments, sharded, local_rank, rank, world)
            if failed:
                if rank == 0:
                    print(f"bench.py: the particle-sharded sweep failed its self-check against the unsharded sweep: {failed}", file=sys.stderr, flush=True)
                raise SystemExit(3)
            self_check = f"sharded sweep of {8192 * world} particles, T=12, bit-identical to the unsharded sweep on all {world} ranks (checked in this run, before the timed region)"
            mark("self-check passed")
        try:
            grp = sharded.make_dist_group(N * world, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn,
                                          pb.basis_fcn, device=f"cuda:{local_rank}")
        except sharded.PgasError as e:   # raised on EVERY rank when any rank failed (allocation, IPC mapping, RCCL communicator)
            print(f"bench.py [rank {rank}]: setup of the particle-sharded sweep failed: {e}", file=sys.stderr, flush=True)
            raise SystemExit(3)
        eng = grp.shards[0].eng
    else:
        eng = pg.cSMC.engine
    if args.chunk >= 0:
        eng.set_option(1, args.chunk)      # PGAS_OPT_PROPAGATE_CHUNK
    if args.local_groups:
        eng.set_option(7, 1)               # PGAS_OPT_LOCAL_GROUPS
    if args.tail_groups:
        eng.set_option(9, 1)               # PGAS_OPT_TAIL_GROUPS
    if args.event_stride > 0:
        eng.set_option(8, args.event_stride)   # PGAS_OPT_EVENT_STRIDE
    if args.max_lead >= 0:
        eng.set_option(11, args.max_lead)      # PGAS_OPT_MAX_LEAD
    if args.no_overlap:
        eng.set_option(3, 0)               # PGAS_OPT_OVERLAP
    if args.prop_lds >= 0:
        eng.set_option(4, args.prop_lds)   # PGAS_OPT_PROPAGATE_LDS

    def one_sweep(sd):
        if grp is not None:
            return sharded.sharded_sweep(grp, sd, ref, A, S)
        return pg.cSMC(sd, ref, A, S)

    mark("sharded group ready" if grp is not None else "engine ready")
    for w in range(args.warmup):
        one_sweep(seed + 1000 + w)
        mark(f"warm-up sweep {w} enqueued")
    barrier()
    mark("warm-up done")
    t0 = time.perf_counter()
    prof_n, prof_ms, prop_n, prop_ms = 0, 0.0, 0, 0.0
    graph_sweeps = 0
    for k in range(args.steps):
        # kernel durations: the LAST timed sweep launches every k_step / k_propagate with start/stop HIP events
        # (hipExtLaunchKernelGGL: the dispatch's own begin/end timestamps, on the streams the kernels run on).  Timing every
        # launch of every sweep would cost ~8 % of the headline value; one sweep of K gives 2 x (T-1) samples inside the timed region.
        timed_sweep = not args.no_profile and (k == args.steps - 1 or prof_all)
        eng.set_profiling((-1 if stride == 1 else stride) if timed_sweep else 0)
        traj_last = one_sweep(seed + k)
        graph_sweeps += 1 if (grp is None and eng.launch_info()["graph"]) else 0
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

    # ---- what was timed must be a sweep: checked on the last timed sweep's traces, after the clock has stopped
    failures = verify_last_sweep(torch, eng, ref, traj_last.reshape(T, -1), owns_conditioned=(not sharded_mode) or rank == world - 1, chase=not sharded_mode)
    if sharded_mode:
        everyone = [None] * world
        dist.all_gather_object(everyone, (failures, traj_last.cpu().numpy().tobytes()))
        failures = [f"rank {r}: {m}" for r, (fl, _) in enumerate(everyone) for m in fl]
        if any(tb != everyone[0][1] for _, tb in everyone):
            failures.append("the ranks disagree on the sampled trajectory")
    if failures:
        if rank == 0:
            print("bench.py: the timed sweeps are INVALID: " + "; ".join(failures), file=sys.stderr, flush=True)
        raise SystemExit(4)
    verified = ("last timed sweep checked after the timed region: trajectory and final log-weights finite, conditioned particle == reference at every t, "
                + ("every rank sampled the same trajectory" if sharded_mode else "trajectory == ancestral path of the final index through the traces"))

    units = N * (T - 1) * args.steps * world
    out = {
        "metric": "particle-steps/sec (N x (T-1) / wall), " + {"smo": "SingleMassOscillator", "vehicle": "Vehicle", "emps": "EMPS"}[args.workload] + " PGAS conditional-SMC sweep",
        "value": units / dt, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic" + (" -- REHEARSAL (PGAS_BENCH_REHEARSE=1): all ranks on one device, gloo collectives; not a measurement" if rehearse else ""),
        "config": {
            "workload": f"{wl_name}, N={N} particles/GPU" + (f" ({N * world} in all)" if sharded_mode else "") + f", T={T}, fp64",
            "particles_per_gpu": N, "particles_total": N * world if sharded_mode else N, "T": T,
            "partition": "particle-sharded" if sharded_mode else ("single GPU" if world == 1 else "replicas"),
            "hip_runtime_version": int(eng.lib.pgas_hip_runtime_version()),
            "launch": ("every sweep enqueued launch by launch" if (sharded_mode or graph_sweeps == 0) else
                       f"{graph_sweeps} of the {args.steps} timed sweeps replay the sweep's HIP graph (PGAS_GRAPH=1; captured once, in the warm-up); the sweep that carries "
                       "the per-launch timing events is enqueued launch by launch"),
            "verified": verified,
            **({"self_check": self_check} if self_check else {}),
            "parallelism": "1 GPU" if world == 1 and not sharded_mode else (
                f"one sweep of {N * world} particles sharded over {world} GPUs (one process per GPU): per time step one RCCL all-gather of the "
                f"segment partials + xGMI peer reads of remote ancestors' rows; per-GPU work is fixed as GPUs are added (weak scaling)"
                if sharded_mode else
                f"{world} independent chains, one per GPU, no data-path collective (replicas; --mode sharded runs the particle-sharded sweep)"),
        },
    }
    if sharded_mode:
        nsegp = grp.shards[0].nsegp
        ag_bytes = 2 * 2 * nsegp * 8   # what one rank contributes per step: (kref, total) x two CDFs x padded segments, 8-byte words
        out["rccl"] = {"rccl_ranks": world, "collectives_per_step": 1, "all_gather_bytes_per_rank_per_step": ag_bytes,
                       "all_gather_bytes_received_per_rank_per_step": ag_bytes * world}
        probe_us = eng.shard_probe_collective(200)
        if probe_us is not None:
            out["rccl"]["all_gather_us_isolated"] = probe_us
            out["rccl"]["note"] = ("all_gather_us_isolated = mean duration of the step's all-gather pair issued back to back 200 times on an idle "
                                   "stream after the timed region (HIP events); inside the sweep it sits on the weight recursion's critical path")
    if rank == 0:
        info = eng.launch_info()
        if prof_n:
            # The two per-step kernels: k_step (search + weights + scans; N particle-steps per launch) and k_propagate (basis, transition, noise;
            # N x chunk particle-steps per launch).  The one with the longer time per step is reported as the dominant kernel; algorithmic bytes of a
            # particle-step = 52, booked in full to it, so `frac` is an upper bound for the pair -- `sweep_frac` is the honest whole-sweep figure.
            us = 1e3 * prof_ms / prof_n
            p_us = 1e3 * prop_ms / max(prop_n, 1)
            p_steps = max(info["chunk"], 1)
            p_us_step = p_us / p_steps
            kname_step = {"local": "k_step<LOCAL>", "tail": "k_step<TAIL>", "k_groups": "k_step (+ k_groups launch)", "small": "k_sweep_small"}[info["groups"]]
            nxv, Dv = pb.nx, len(pb.basis_fcn.sel)
            kname_prop = f"k_propagate<{nxv},{Dv},{info['JP']},{info['P']}>"
            per_launch = ALG_BYTES_PER_PARTICLE_STEP * N
            dom_is_step = us >= p_us_step
            dom = dict(kernel=kname_step, avg_launch_us=us, launches_timed=prof_n, steps_per_launch=1) if dom_is_step else \
                dict(kernel=kname_prop, avg_launch_us=p_us, launches_timed=prop_n, steps_per_launch=p_steps)
            oth = dict(kernel=kname_prop, avg_launch_us=p_us, launches_timed=prop_n, steps_per_launch=p_steps) if dom_is_step else \
                dict(kernel=kname_step, avg_launch_us=us, launches_timed=prof_n, steps_per_launch=1)
            achieved = per_launch * dom["steps_per_launch"] / (dom["avg_launch_us"] * 1e-6) / 1e9
            oth["achieved_GBs"] = per_launch * oth["steps_per_launch"] / (oth["avg_launch_us"] * 1e-6) / 1e9
            sweep_GBs = ALG_BYTES_PER_PARTICLE_STEP * units / dt / world / 1e9   # per GPU
            traffic, traffic_src = None, None
            for tf in (("traffic_r03.json", "traffic_r02.json", "traffic_r01.json") if args.workload == "smo" else ()):   # the stored counters are of the smo workload
                tp = os.path.join(ROOT, "profiles", tf)
                if os.path.exists(tp):
                    tj = json.load(open(tp))
                    key = "k_step_hbm_bytes_per_launch" if dom_is_step else "k_propagate_hbm_bytes_per_step"
                    if tj.get(key) is not None:
                        traffic = tj[key] * (1 if dom_is_step else dom["steps_per_launch"])
                        traffic_src = f"profiles/{tf} ({tj.get('collected', 'date unknown')}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, " \
                                      f"FETCH_SIZE x 2 per the gfx950 correction; a stored measurement, not taken in this run"
                        oth["traffic_per_step"] = tj.get("k_propagate_hbm_bytes_per_step" if dom_is_step else "k_step_hbm_bytes_per_launch")
                        if "valu" in tj:
                            out["valu"] = tj["valu"]
                            tot = sum((v.get("valu_instr_per_particle_step") or 0.0) for v in tj["valu"].values())
                            # every VALU instruction of a wave64 occupies its SIMD for 4 cycles (16 lanes per clock); 1024 SIMDs at ~2.1 GHz under this load
                            out["valu"]["summary"] = {"lane_instr_per_particle_step": tot, "issue_bound_us_per_step": tot * N / 64 * 4 / 1024 / 2.1e3,
                                                      "measured_us_per_step": 1e6 * dt / args.steps / (T - 1),
                                                      "note": f"stored PMC counts (SQ_INSTS_VALU x 64 / N per kernel, profiles/{tf}), not taken in this run"}
                        break
            # SURVEY 8(d), secondary figure: algorithmic flops per particle-step = 2 M nx (Phi A^T) + M (D - 1) (products) against the
            # fp64 vector peak -- the bound that matters for the M = 729 configurations
            flops_ps = 2 * eng.M * nxv + eng.M * (Dv - 1)
            out["valu_roofline"] = {
                "flops_per_particle_step": flops_ps, "achieved": flops_ps * units / dt / world / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": flops_ps * units / dt / world / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                "note": "whole sweep; flops = 2 M nx + M (D - 1) per particle-step (SURVEY.md section 8d), sines, random numbers and the weight recursion not counted",
            }
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src, "kernel": dom["kernel"], "avg_launch_us": dom["avg_launch_us"],
                "launches_timed": dom["launches_timed"], "steps_per_launch": dom["steps_per_launch"],
                "launch_sampling": f"every {'launch' if stride == 1 else str(stride) + 'th launch'} of {'every timed sweep' if prof_all else 'the last timed sweep'} "
                                   "carries start/stop HIP events (hipExtLaunchKernelGGL)",
                "alg_bytes_per_launch": per_launch * dom["steps_per_launch"],
                "sweep_achieved": sweep_GBs, "sweep_frac": sweep_GBs / HBM_PEAK_GBS,
                "sweep_note": "sweep_* = 52 B x N x (T-1) / wall of the whole sweep per GPU: both kernels, launch gaps, final draw and back-trace included",
                "hbm_copy_GBs": hbm_copy_rate(torch, eng.device),   # measured attainable rate of a 1 GiB device copy, outside the timed region
                "second_kernel": oth,
                "note": "the two kernels run concurrently on two streams; the period of a step is the weight recursion's dependent chain (k_step + k_groups + "
                        "two launch boundaries), with k_propagate hidden under it.  Together they issue ~850 vector instructions per particle-step (~26 us per "
                        "step at the nominal 4 cycles each, ~31 us at the 5.5 a stream of independent v_fma_f64 really reaches: "
                        "profiles/r03_probe_valu_f64_rate.txt) and move 114 MB of real traffic per step against 54.5 MB algorithmic: DESIGN.md sections 5 and 8",
            }
        if args.cpu_steps > 0 and world == 1:
            Ah, Sh = A.cpu().numpy(), S.cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(pb, Ah, Sh, N, seed, args.cpu_steps)
            if args.workload == "smo":
                out["cpu_baseline"]["config1"] = numpy_baseline_config1(pb, Ah, Sh, seed)
                # the engine at the same size (N = 200: the whole sweep is one launch of one workgroup, k_sweep_small), beside the CPU figures
                small = pgas_amd.condSequentialMonteCarlo(200, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn,
                                                          pb.basis_fcn, device=f"cuda:{local_rank}")
                small(seed, ref, A, S)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for k in range(5):
                    small(seed + 1 + k, ref, A, S)
                torch.cuda.synchronize()
                ms = 1e3 * (time.perf_counter() - t1) / 5
                out["cpu_baseline"]["config1"]["engine"] = {"value": 200 * (T - 1) / (ms * 1e-3), "unit": "particle-steps/s", "ms_per_sweep": ms,
                                                            "kernel": "k_sweep_small" if small.engine.launch_info()["small"] else "general path",
                                                            "note": "this repository's engine on the same N = 200, T = 2000 sweep (1 x MI355X, mean of 5 sweeps)"}
                small.engine.close()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
        if not rehearse:
            import faulthandler
            faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
