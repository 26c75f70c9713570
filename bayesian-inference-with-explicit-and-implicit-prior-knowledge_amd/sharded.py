"""Particle-sharded conditional-SMC sweep.

The reference is a single process; this is the build's multi-GPU extension of
``condSequentialMonteCarlo.__call__`` (src/PGAS.py:176-228).  Rank r owns particles
[r N/G, (r+1) N/G); the conditioned particle N-1 lives on the last rank.  Per time step there is ONE
collective: an all-gather of the per-segment softmax partials (2 x 2 x nseg_local 8-byte words per rank),
after which every rank computes the identical group records (k_groups), so ancestors -- and the sampled
trajectory -- do not depend on the number of ranks.  Ancestors that live on another rank are read
through xGMI peer mappings of the scan buffers.

Two groups implement the exchange:
  * ``LocalGroup``  -- several shards in ONE process on one device (tests, single-GPU emulation);
  * ``DistGroup``   -- one shard per process, torch.distributed (backend "nccl" = RCCL) + HIP IPC handles; the time loop
                       runs inside the library (pgas_shard_sweep).
"""
from __future__ import annotations

import numpy as np
import torch

from . import random as prng
from ._lib import Engine, PgasError

PH_INIT, PH_PROPAGATE, PH_STEP, PH_GROUPS, PH_FINAL_SCAN, PH_FINAL, PH_BACKTRACE = range(7)


def shard_layout(N_global: int, world: int, seg: int = 1024):
    """Particles per rank; every rank gets the same whole number of segments."""
    if N_global % (world * seg):
        raise ValueError(f"N = {N_global} must be a multiple of world * {seg} = {world * seg}")
    return N_global // world


class _Shard:
    """One rank's engine plus torch views of the buffers that take part in the exchange."""

    def __init__(self, eng: Engine, rank: int, world: int):
        self.eng = eng
        eng.shard_setup(rank, world)
        self.ptrs, (self.nsegp, self.Nl, self.T) = eng.shard_buffers()
        w = 2 * self.nsegp
        self.segm_w = [eng.dev_tensor(self.ptrs[9 + i], (w,), torch.float64) for i in range(2)]
        self.segs_w = [eng.dev_tensor(self.ptrs[11 + i], (w,), torch.int64) for i in range(2)]
        self.segm_g = [eng.dev_tensor(self.ptrs[13 + i], (world * w,), torch.float64) for i in range(2)]
        self.segs_g = [eng.dev_tensor(self.ptrs[15 + i], (world * w,), torch.int64) for i in range(2)]
        # the seven buffers peers read (two segment cumsums, la, h, ln, x, anc), block by block: the traces are row blocks of <= 1 GiB
        self.nblk = [eng.shard_layout(k)[0] for k in range(7)]

    def blocks(self):
        """[(which, blk, device pointer)] of every peer-visible block of this rank."""
        return [(k, b, self.eng.shard_block(k, b)[0]) for k in range(7) for b in range(self.nblk[k])]


class LocalGroup:
    """All shards live in this process (same device): pointers are shared directly, the all-gather is a copy."""

    def __init__(self, shards):
        self.shards = shards
        for s in shards:
            for peer, o in enumerate(shards):
                for which, blk, ptr in o.blocks():
                    s.eng.shard_set_peer_block(peer, which, blk, ptr)

    def all_gather(self, parity):
        w = 2 * self.shards[0].nsegp
        for s in self.shards:
            for r, o in enumerate(self.shards):
                s.segm_g[parity][r * w:(r + 1) * w].copy_(o.segm_w[parity])
                s.segs_g[parity][r * w:(r + 1) * w].copy_(o.segs_w[parity])

    def barrier(self):
        torch.cuda.synchronize()


def agree_on(dist, group, step, fn):
    """Run the local part `fn` of a setup step and agree on its outcome across the ranks of `group`: a failure on one rank raises
    PgasError on every rank instead of leaving the others waiting in the next collective."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    err = None
    try:
        fn()
    except Exception as e:   # noqa: BLE001  (reported below, on every rank)
        err = f"rank {rank}: {type(e).__name__}: {e}"
    errs = [None] * world
    dist.all_gather_object(errs, err, group=group)
    bad = [e for e in errs if e]
    if bad:
        raise PgasError(f"sharded sweep setup failed at '{step}': " + "; ".join(bad))


class DistGroup:
    """One shard per process.  The time loop runs inside the library (pgas_shard_sweep) on every backend:
      * "nccl": the per-step all-gather is an RCCL call on the sweep's stream (own communicator; torch.distributed only carries
        the 128-byte id and, once, the IPC handles);
      * "gloo": the library calls back into `_host_all_gather`, which stages the partials through the host -- the development /
        test path for several ranks on ONE device, where RCCL refuses duplicate devices.  Same loop, same launches."""

    def __init__(self, shard, group=None):
        import torch.distributed as dist

        self.dist, self.group, self.shards = dist, group, [shard]
        self.library_loop = True
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)

        def agreed(step, fn):
            agree_on(dist, group, step, fn)

        # Every peer-visible block crosses the process boundary as one HIP IPC handle.  The blocks are at most 1 GiB (pgas_shard_setup):
        # the HIP 7.0 runtime PyTorch 2.10+rocm7.0 bundles never returns from hipIpcOpenMemHandle for an allocation of 2 GiB or more
        # (2047 MiB opens in 0.4 ms), and pgas_ipc_export refuses such a block instead of letting a peer hang on it.
        handles = []
        agreed("export the IPC handles", lambda: handles.extend((k, b, shard.eng.ipc_export(k, b)) for k, b, _ in shard.blocks()))
        everyone = [None] * world
        dist.all_gather_object(everyone, handles, group=group)

        def open_peers():
            for peer, hs in enumerate(everyone):
                if peer == rank:
                    for which, blk, ptr in shard.blocks():
                        shard.eng.shard_set_peer_block(peer, which, blk, ptr)
                else:
                    for which, blk, h in hs:
                        shard.eng.shard_set_peer_block(peer, which, blk, shard.eng.ipc_open(h))

        agreed("open the peers' IPC handles", open_peers)
        if self.backend == "nccl":
            ident = [None]

            def make_id():
                if rank == 0:
                    ident[0] = shard.eng.shard_unique_id()

            agreed("RCCL unique id", make_id)
            dist.broadcast_object_list(ident, src=0, group=group)
            agreed("RCCL communicator", lambda: shard.eng.shard_comm_init(ident[0]))
        else:
            shard.eng.shard_set_collective(self._host_all_gather)

    def _host_all_gather(self, parity, stream_ptr):
        s = self.shards[0]
        # the library hands over the raw hipStream_t its loop runs on; NULL (None through ctypes) is the device's default stream
        st = torch.cuda.ExternalStream(int(stream_ptr), device=s.eng.device) if stream_ptr else torch.cuda.default_stream(s.eng.device)
        st.synchronize()
        if parity < 0:   # end-of-sweep barrier: nobody rewrites traces its peers are still chasing ancestors through
            self.dist.barrier(group=self.group)
            return 0
        world = self.dist.get_world_size(self.group)
        with torch.cuda.stream(st):
            for dst, src in ((s.segm_g[parity], s.segm_w[parity]), (s.segs_g[parity], s.segs_w[parity])):
                parts = [torch.empty(src.shape, dtype=src.dtype) for _ in range(world)]
                self.dist.all_gather(parts, src.cpu(), group=self.group)
                dst.copy_(torch.cat(parts))
        return 0

    def barrier(self):
        torch.cuda.synchronize()
        self.dist.barrier(group=self.group)


def sharded_sweep(group, seed, ref, coeff_mat, error_cov, propagate_chunk=0):
    """Run one sweep on every shard of `group`; returns the trajectory (T, nx) (identical on every rank)."""
    shards = group.shards
    T = shards[0].T
    seed = prng.as_key(seed)
    refs, trajs = [], []
    for s in shards:
        s.eng.set_params(coeff_mat, error_cov)
        r = ref if isinstance(ref, torch.Tensor) else torch.as_tensor(np.asarray(ref, dtype=np.float64))
        refs.append(r.to(device=s.eng.device, dtype=torch.float64).reshape(T, s.eng.nx).contiguous())
        trajs.append(torch.empty((T, s.eng.nx), dtype=torch.float64, device=s.eng.device))
    if getattr(group, "library_loop", False):
        shards[0].eng.shard_sweep(seed, refs[0], trajs[0], propagate_chunk)   # pgas_shard_sweep: loop + collective inside the library
        return trajs[0]
    # several shards in one process: the same phases, interleaved over the shards by hand
    chunk = propagate_chunk if propagate_chunk > 0 else T
    for s, r in zip(shards, refs):
        s.eng.shard_run(PH_INIT, seed=seed, ref=r)
        for t0 in range(1, T, chunk):
            s.eng.shard_run(PH_PROPAGATE, t0, min(t0 + chunk, T), seed=seed, ref=r)
    for t in range(1, T + 1):
        for s in shards:
            s.eng.shard_run(PH_STEP, t, seed=seed)
        if t < T:
            group.all_gather(t & 1)          # the one collective of the step
            for s in shards:
                s.eng.shard_run(PH_GROUPS, t, seed=seed)
    if T == 1:
        for s in shards:
            X, A, LW, _ = s.eng.traces()
            LW.zero_()
    for s in shards:
        s.eng.shard_run(PH_FINAL_SCAN, seed=seed)
    group.all_gather(T & 1)
    for s, tr in zip(shards, trajs):
        s.eng.shard_run(PH_FINAL, seed=seed)
        s.eng.shard_run(PH_BACKTRACE, seed=seed, traj=tr)
    group.barrier()                          # peers may still be reading this rank's traces
    return trajs if len(trajs) > 1 else trajs[0]


def _shard_engine(Nl, args, device, trace_block_bytes):
    eng = Engine(Nl, *args, device=device)
    if trace_block_bytes:   # test knob (PGAS_OPT_TRACE_BLOCK_BYTES): block boundaries inside short sweeps; default = 1 GiB blocks
        eng.set_option(12, int(trace_block_bytes))
    return eng


def make_local_group(world, N_global, observations, inputs, init_state_mean, init_state_cov, likelihood_fcn, basis_fcn, device=None, trace_block_bytes=None):
    Nl = shard_layout(N_global, world)
    args = (observations, inputs, init_state_mean, init_state_cov, likelihood_fcn, basis_fcn)
    shards = [_Shard(_shard_engine(Nl, args, device, trace_block_bytes), r, world) for r in range(world)]
    return LocalGroup(shards)


def make_dist_group(N_global, observations, inputs, init_state_mean, init_state_cov, likelihood_fcn, basis_fcn, device=None, group=None, trace_block_bytes=None):
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    Nl = shard_layout(N_global, world)
    made = []   # the engine (device tables, scan buffers) is created inside an agreed step too: a rank that cannot allocate takes the others down with a message
    agree_on(dist, group, "create this rank's engine", lambda: made.append(
        _Shard(_shard_engine(Nl, (observations, inputs, init_state_mean, init_state_cov, likelihood_fcn, basis_fcn), device, trace_block_bytes), rank, world)))
    return DistGroup(made[0], group)
