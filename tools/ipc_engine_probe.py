"""Two processes on one device exchange the IPC handles of their shard buffers and time every pgas_ipc_open (development aid: where does
the setup of a large particle-sharded sweep spend its time?).  usage: ipc_engine_probe.py [N_local] [T]"""
import os, sys, time
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, Nl, T, qs):
    sys.path.insert(0, ROOT)
    import torch
    import pgas_amd
    from pgas_amd import experiments
    from pgas_amd._lib import Engine
    pb = experiments.smo_pgas(T=T)
    t0 = time.perf_counter()
    eng = Engine(Nl, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn, device="cuda:0")
    eng.shard_setup(rank, 2)
    ptrs, sz = eng.shard_buffers()
    torch.cuda.synchronize()
    print(f"[{rank}] engine + shard_setup {time.perf_counter() - t0:.2f} s, sizes {sz}", flush=True)
    handles = [eng.ipc_export(k) for k in range(7)]
    qs[1 - rank].put(handles)
    theirs = qs[rank].get(timeout=120)
    names = ["c1[0]", "c1[1]", "la", "h", "ln", "x", "anc"]
    if os.environ.get("SERIAL") == "1" and rank == 1:
        qs[rank].get(timeout=300)          # rank 0 maps first; this rank sits outside the HIP runtime meanwhile
    for k, h in enumerate(theirs):
        t0 = time.perf_counter()
        p = eng.ipc_open(h)
        print(f"[{rank}] ipc_open {names[k]:6s} {1e3 * (time.perf_counter() - t0):9.1f} ms -> {p:#x}", flush=True)
    qs[1 - rank].put("done")
    if not (os.environ.get("SERIAL") == "1" and rank == 1):
        qs[rank].get(timeout=300)


if __name__ == "__main__":
    Nl = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    ctx = mp.get_context("spawn")
    qs = [ctx.Queue(), ctx.Queue()]
    ps = [ctx.Process(target=worker, args=(r, Nl, T, qs)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(int(os.environ.get("JOIN_S", "240")))
        if p.is_alive():
            print("a worker is still running in time: terminating", flush=True)
            p.terminate()
