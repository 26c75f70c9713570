#!/bin/bash
# Evidence for the marginalised family: kernel timings, rocprofv3 kernel stats, the SMO driver's log.
set -o pipefail
R=${1:-r01}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python tools/marginal_times.py 200 16384 131072 1048576 2>&1 | grep -v amdgpu.ids > $O/marginal_times.txt || exit 1
cat $O/marginal_times.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mstats -- python3 tools/marginal_times.py 1048576 > $O/marginal_under_rocprof.txt 2>&1 || { tail $O/marginal_under_rocprof.txt; exit 1; }
cp $(ls $O/mstats/*/*kernel_stats.csv | head -1) $O/marginal_kernel_stats.csv && head -12 $O/marginal_kernel_stats.csv | cut -c1-220
rm -rf $O/mstats
timeout -k 10 900 python examples/SingleMassOscillator_Simulation.py --iterations ${2:-60} --out $O/SingleMassOscillator.mat 2>&1 | grep -v amdgpu.ids > $O/smo_driver.log || { tail $O/smo_driver.log; exit 1; }
cat $O/smo_driver.log
rm -f $O/SingleMassOscillator.mat
