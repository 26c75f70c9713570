"""Multi-PROCESS particle-sharded sweep on the GPU box.

* two ranks (processes) share the one MI355X, exchange HIP IPC handles of their buffers and read each other's memory through
  the peer mappings; the time loop is the library's (pgas_shard_sweep), its per-step all-gather staged over `gloo` through the
  collective callback (RCCL refuses two ranks on one device);
* the same loop with the RCCL all-gather, one rank (all a single device admits);
* with >= 2 visible GPUs: two ranks, one per GPU, backend "nccl" -- the configuration bench.py --gpus 2 runs (skipped on a
  one-GPU box).
The trajectory and the traces must be the single-device / oracle ones bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from common import ROOT, canon_model, experiments

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, q, backend="gloo", one_gpu_per_rank=False):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist

    dev = rank if one_gpu_per_rank else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pgas_amd  # noqa: F401
        from pgas_amd import sharded

        pb = experiments.smo_pgas(T=12)
        A, S = experiments.initial_params(pb)
        # traces in row blocks of 128 KiB (default 1 GiB): every peer-visible trace crosses the process boundary as SEVERAL IPC handles
        grp = sharded.make_dist_group(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn,
                                      device=f"cuda:{dev}", trace_block_bytes=1 << 17)
        assert grp.shards[0].nblk[5] > 1
        assert grp.library_loop and grp.backend == backend
        traj = sharded.sharded_sweep(grp, 12345678, pb.X_true, A, S, propagate_chunk=4)
        traj = sharded.sharded_sweep(grp, 12345678, pb.X_true, A, S, propagate_chunk=4)   # twice: the end-of-sweep collective orders the reuse
        X, ANC, LW, _ = grp.shards[0].eng.traces()
        q.put((rank, traj.cpu().numpy(), ANC[: pb.T - 1].cpu().numpy()))
        dist.barrier()
    except Exception as exc:   # report instead of letting the parent wait for its queue timeout
        import traceback

        q.put((rank, "error", "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))))
    finally:
        dist.destroy_process_group()


def _collect(q, procs, timeout):
    res = []
    try:
        for _ in procs:
            r = q.get(timeout=timeout)
            if isinstance(r[1], str):
                raise AssertionError(f"rank {r[0]} failed:\n{r[2]}")
            res.append(r)
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:   # never leave a rank behind (it would keep the GPU and the pytest process alive)
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
    return sorted(res, key=lambda r: r[0])


def test_two_process_sharded_sweep_matches_oracle():
    world, N = 2, 8192
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = _collect(q, procs, 300)
    pb = experiments.smo_pgas(T=12)
    A, S = experiments.initial_params(pb)
    cm = canon_model(pb, N)
    LS, LSinv, cS = cm.chol_parts(S)
    trajo, Xo, ANCo, lwo = cm.sweep(12345678, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov))
    Nl = N // world
    for rank, traj, anc in res:
        assert np.array_equal(traj, trajo), f"rank {rank}: trajectory differs"
        assert np.array_equal(anc, ANCo[:, rank * Nl:(rank + 1) * Nl]), f"rank {rank}: ancestor trace differs"


def test_library_loop_with_rccl_single_rank():
    """pgas_shard_sweep (time loop + RCCL all-gather inside the library) with backend "nccl".  One MI355X admits one RCCL rank, so
    this exercises the communicator set-up, the stream-ordered collectives and the phase sequence with world = 1; the two-rank
    decomposition itself is covered by the gloo-staged test above and by the in-process shard tests of test_gpu_parity.py."""
    world, N = 1, 4096
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, world, _free_port(), N, q, "nccl"))
    p.start()
    ((rank, traj, anc),) = _collect(q, [p], 300)
    pb = experiments.smo_pgas(T=12)
    A, S = experiments.initial_params(pb)
    cm = canon_model(pb, N)
    LS, LSinv, cS = cm.chol_parts(S)
    trajo, Xo, ANCo, lwo = cm.sweep(12345678, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov))
    assert np.array_equal(traj, trajo) and np.array_equal(anc, ANCo)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one RCCL rank per device")
def test_two_gpu_rccl_sharded_sweep_matches_oracle():
    """The multi-GPU configuration itself: two processes, one GPU each, backend "nccl" (= RCCL over xGMI), peers' cumsums and
    log-likelihood rows read through IPC peer mappings.  Bit-exact against the oracle like every other partition."""
    world, N = 2, 8192
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, q, "nccl", True)) for r in range(world)]
    for p in procs:
        p.start()
    res = _collect(q, procs, 600)
    pb = experiments.smo_pgas(T=12)
    A, S = experiments.initial_params(pb)
    cm = canon_model(pb, N)
    LS, LSinv, cS = cm.chol_parts(S)
    trajo, Xo, ANCo, lwo = cm.sweep(12345678, pb.X_true, A, LS, LSinv, cS, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov))
    Nl = N // world
    for rank, traj, anc in res:
        assert np.array_equal(traj, trajo), f"rank {rank}: trajectory differs"
        assert np.array_equal(anc, ANCo[:, rank * Nl:(rank + 1) * Nl]), f"rank {rank}: ancestor trace differs"
