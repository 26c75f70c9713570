"""world_size-2 (and 4) `gloo` tests of the particle-sharded decomposition on CPU.

The GPU kernels cannot run here, so the per-rank arithmetic is done by the canonical oracle (test infrastructure);
what is under test is the N>1 data flow the product uses: every rank scans only ITS segments, the per-segment partials are
all-gathered with torch.distributed, every rank then evaluates the cross-segment scan on the gathered arrays and resamples
only ITS slots -- and the result must equal the single-rank result exactly, whatever the number of ranks.
Also covers the host-side layout helper of pgas_amd.sharded.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import ROOT, canon, canon_model, experiments


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgas_amd.sharded import shard_layout

        pb = experiments.smo_pgas(T=6)
        A, S = experiments.initial_params(pb)
        cm = canon_model(pb, N)
        LS, LSinv, cS = cm.chol_parts(S)
        seed = 4242
        x0 = cm.init_state(seed, pb.init_state_mean, np.linalg.cholesky(pb.init_state_cov), pb.X_true[0])
        lw_prev = None
        x = x0
        for t in (1, 2, 3):
            lw, xn, anc, dbg = cm.step(t, seed, x, lw_prev, A, LS, LSinv, cS, pb.X_true[t], debug=True)  # single-rank truth
            Nl = shard_layout(N, world)
            sl = slice(rank * Nl, (rank + 1) * Nl)
            # this rank's segments only
            segm_l, segs_l, c_l = canon.segment_partials(dbg["lw1"][sl])
            # ONE collective on the partials (+ the cumsums, which the GPU build reads through peer mappings instead)
            gm = [torch.empty(len(segm_l), dtype=torch.float64) for _ in range(world)]
            gs = [torch.empty(len(segs_l), dtype=torch.int64) for _ in range(world)]
            gc = [torch.empty(len(c_l), dtype=torch.int64) for _ in range(world)]
            dist.all_gather(gm, torch.from_numpy(segm_l))
            dist.all_gather(gs, torch.from_numpy(segs_l.view(np.int64)))
            dist.all_gather(gc, torch.from_numpy(c_l.view(np.int64)))
            segm = torch.cat(gm).numpy()
            segs = torch.cat(gs).numpy().view(np.uint64)
            c = torch.cat(gc).numpy().view(np.uint64)
            mine = canon.resample_range(segm, segs, c, N, dbg["u"][0], rank * Nl, (rank + 1) * Nl)
            expect = anc[sl].copy()
            if rank == world - 1:
                mine[-1] = expect[-1]  # the conditioned slot takes the separately drawn ancestor (src/PGAS.py:127)
            if not np.array_equal(mine, expect):
                q.put((rank, t, int((mine != expect).sum())))
                return
            lw_prev, x = lw, xn
        q.put((rank, "ok", 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 4096), (4, 8192), (2, 2048)])
def test_sharded_resampling_equals_single_rank(world, N):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_shard_layout():
    from pgas_amd.sharded import shard_layout

    assert shard_layout(1 << 23, 8) == 1 << 20
    assert shard_layout(4096, 2) == 2048
    with pytest.raises(ValueError):
        shard_layout(5000, 2)
    with pytest.raises(ValueError):
        shard_layout(3072, 2)


def _agree_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgas_amd.sharded import PgasError, agree_on

        agree_on(dist, None, "a step every rank completes", lambda: None)   # no error anywhere: returns on every rank

        def local_part():
            if rank == 1:
                raise RuntimeError("cannot map the peer's buffer")

        try:
            agree_on(dist, None, "open the peers' IPC handles", local_part)
            q.put((rank, "no error raised"))
        except PgasError as e:
            q.put((rank, str(e)))
        dist.barrier()   # every rank is still in step with the others afterwards
    finally:
        dist.destroy_process_group()


def test_setup_failure_on_one_rank_raises_on_every_rank():
    """pgas_amd.sharded.agree_on: what keeps a multi-GPU run from hanging when one rank cannot open a peer's IPC handle or build
    its RCCL communicator -- every rank gets the same PgasError (bench.py then falls back to independent chains)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = dict(q.get(timeout=120) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert set(got) == {0, 1}
    for r in (0, 1):
        assert "open the peers' IPC handles" in got[r] and "rank 1: RuntimeError: cannot map the peer's buffer" in got[r], got
