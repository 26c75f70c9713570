#!/bin/bash
# Multi-rank code path of bench.py on ONE GPU (PGAS_BENCH_REHEARSE=1: every rank on device 0, gloo collectives through the library's
# host-callback all-gather, real HIP IPC mappings between the processes, the process's own HIP runtime -- no LD_PRELOAD, no exec).
# usage: tools/rehearse.sh RANKS PARTICLES_PER_RANK T OUT_PREFIX
set -o pipefail
R=${1:-2}; N=${2:-1048576}; T=${3:-2000}; OUT=${4:-gpurun_out/rehearse_${R}}
export PGAS_BENCH_REHEARSE=1
timeout -k 10 ${REHEARSE_TIMEOUT:-420} python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port $((29500 + R)) \
    bench.py --gpus $R --steps 2 --warmup 1 --particles $N --T $T --cpu-steps 0 > $OUT.json 2> $OUT.err
rc=$?
echo "rehearsal ranks=$R N=$N T=$T rc=$rc"
tail -c 600 $OUT.json
[ $rc -eq 0 ] || tail -40 $OUT.err
exit $rc
