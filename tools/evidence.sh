#!/bin/bash
# Round evidence: GPU tests, default bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE passes.
# usage (on the GPU box): tools/evidence.sh r01
set -o pipefail
R=${1:-r01}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
cat $O/bench_default.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --cpu-steps 0 > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err || { tail $O/rocprof_stats.err; exit 1; }
cat $O/bench_under_rocprof.json
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv && head -8 $O/kernel_stats.csv
rm -f $O/stats/*/*kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-profile --T 200 > /dev/null 2> $O/pmc_$c.err || { tail $O/pmc_$c.err; exit 1; }
  python3 tools/pmc_kernels.py $O/pmc_$c raw > $O/pmc_$c.txt && cat $O/pmc_$c.txt
  rm -rf $O/pmc_$c
done
timeout -k 10 300 python tools/config_times.py > $O/config_times.txt 2>&1 && cat $O/config_times.txt
