# bench.py's own handling of the >= 2 GiB hipIpcOpenMemHandle hang of PyTorch's bundled HIP runtime (one-GPU rehearsal, two ranks)
mkdir -p gpurun_out
echo "== 2-rank rehearsal, T=300: bench re-execs itself on the system runtime"
PGAS_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-steps 0 --T 300 2>gpurun_out/pl2.err | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['n_gpus'], round(j['ms_per_step'],1), j['config']['partition'], j['config'].get('hip_runtime','-')[:60], j['config'].get('fallback_reason','-')[:200])"
echo "== the same with PGAS_NO_PRELOAD=1: refused on every rank, independent chains instead"
PGAS_NO_PRELOAD=1 PGAS_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-steps 0 --T 300 2>gpurun_out/pl3.err | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['n_gpus'], round(j['ms_per_step'],1), j['config']['partition'], j['config'].get('hip_runtime','-')[:60], j['config'].get('fallback_reason','-')[:300])"
grep "falling back" gpurun_out/pl3.err | cut -c1-300
echo "== default single-GPU bench (untouched path)"
timeout -k 10 200 python bench.py --cpu-steps 0 --steps 3 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['n_gpus'], round(j['ms_per_step'],2), j['config']['partition'], j['config'].get('hip_runtime','-'))"
