#!/bin/bash
# Copy the summaries the judge reads from gpurun_out/<round> (scratch) into profiles/ (tracked).  usage: tools/collect_profiles.sh r02
R=${1:-r02}; O=gpurun_out/$R
set -e
cp $O/bench_default.json profiles/${R}_bench_default.json
cp $O/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp $O/bench_emps.json profiles/${R}_bench_emps.json
cp $O/bench_vehicle.json profiles/${R}_bench_vehicle.json
cp $O/kernel_stats.csv profiles/${R}_kernel_stats.csv
cp $O/pmc_FETCH_SIZE.txt profiles/${R}_pmc_fetch_size.txt
cp $O/pmc_WRITE_SIZE.txt profiles/${R}_pmc_write_size.txt
for n in insts cycles grbm lds; do cp $O/pmc_sq_$n.txt profiles/${R}_pmc_sq_$n.txt; done
cp $O/config_times.txt profiles/${R}_config_times.txt
cp $O/traffic_$R.json profiles/traffic_$R.json
for f in syrk_time.txt syrk_kernel_stats.csv syrk_pmc.txt; do [ -f $O/$f ] && cp $O/$f profiles/${R}_$f; done
tail -3 $O/pytest_gpu.log > profiles/${R}_pytest_gpu_tail.txt
ls profiles/ | grep $R
