"""Per-step time of the particle-sharded sweep's in-library loop (pgas_shard_sweep, RCCL) with ONE rank (development aid): what a
rank of the 8-GPU configuration executes per step, minus the wire time of the collective."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import pgas_amd
from pgas_amd import experiments, sharded
N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20, 200
pb = experiments.smo_pgas(T=T)
A, S = experiments.initial_params(pb)
grp = sharded.make_dist_group(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn, device="cuda:0")
for chunk in (0, 16, 1):
    sharded.sharded_sweep(grp, 1, pb.X_true, A, S, propagate_chunk=chunk); torch.cuda.synchronize()
    t0 = time.perf_counter(); sharded.sharded_sweep(grp, 2, pb.X_true, A, S, propagate_chunk=chunk); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"library loop, N={N}, T={T}, propagate chunk {chunk or T}: {1e3*dt:.2f} ms/sweep = {1e6*dt/(T-1):.1f} us/step = {N*(T-1)/dt:.3e} particle-steps/s", flush=True)
dist.destroy_process_group()
