#!/bin/bash
# Round evidence: GPU tests, default bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE passes.
# usage (on the GPU box): tools/evidence.sh r02
set -o pipefail
R=${1:-r02}
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
for w in emps vehicle; do timeout -k 10 300 python bench.py --workload $w --cpu-steps 0 > $O/bench_$w.json 2> $O/bench_$w.err && cat $O/bench_$w.json; done
tools/pmc_sq.sh $R
python3 tools/make_traffic.py $O $O/traffic_$R.json > /dev/null && echo traffic json written

# pgas_suffstats at M = 729, T = 2000 (EMPS): per-kernel durations and the f64 MFMA counters of k_syrk_lds
ONLY=EMPS timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sy -- python3 tools/syrk_time.py 0 2>/dev/null | grep 'pgas_suffstats call' > $O/syrk_time.txt
grep -h "Name\|k_syrk\|k_traj" $O/sy/*/*kernel_stats.csv > $O/syrk_kernel_stats.csv; rm -rf $O/sy; cat $O/syrk_time.txt $O/syrk_kernel_stats.csv
ONLY=EMPS timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/sp -- python3 tools/syrk_time.py 0 > /dev/null 2> $O/syrk_pmc.err
python3 tools/pmc_kernels.py $O/sp raw | grep "syrk\|traj" > $O/syrk_pmc.txt; rm -rf $O/sp; cat $O/syrk_pmc.txt
