#!/bin/bash
# one-GPU rehearsal of what the driver runs at N>1: the sharded bench code path with a single RCCL rank, next to the unsharded sweep
set -o pipefail
mkdir -p gpurun_out
python bench.py --cpu-steps 0 > gpurun_out/bench_unsharded.json 2> gpurun_out/bench_unsharded.err || exit 1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --mode sharded --cpu-steps 0 \
    > gpurun_out/bench_sharded_w1.json 2> gpurun_out/bench_sharded_w1.err || { tail -20 gpurun_out/bench_sharded_w1.err; exit 2; }
python - <<'PY'
import json
for f in ("bench_unsharded", "bench_sharded_w1"):
    j = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
    r = j.get("roofline", {})
    print(f, "ms/sweep %.2f" % j["ms_per_step"], "dominant", r.get("kernel"), "%.1f us" % r.get("avg_launch_us", 0), "second", r.get("second_kernel", {}).get("kernel"),
          "%.1f us" % r.get("second_kernel", {}).get("avg_launch_us", 0), "sweep_frac %.3f" % r.get("sweep_frac", 0), j.get("rccl"))
PY
