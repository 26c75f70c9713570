"""Host enqueue time of the library's sharded time loop (k_propagate / k_step / RCCL all-gather / k_groups per step) against its
device time, one RCCL rank (development aid).  Run under: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 ..."""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgas_amd
from pgas_amd import experiments, sharded
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
N, T = 1 << 20, 2000
pb = experiments.smo_pgas(T=T)
A, S = experiments.initial_params(pb)
grp = sharded.make_dist_group(N, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn, device="cuda:0")
ref = torch.as_tensor(pb.X_true, device="cuda:0")
sharded.sharded_sweep(grp, 1, ref, A, S); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    sharded.sharded_sweep(grp, 2 + rep, ref, A, S)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"sharded (1 rank): enqueue {1e3 * (t1 - t0):.1f} ms, until done {1e3 * (t2 - t0):.1f} ms", flush=True)
dist.destroy_process_group()
