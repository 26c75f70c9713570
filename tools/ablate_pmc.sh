#!/bin/bash
# Dynamic instruction counts of the sweep kernel for each ablated build (timing experiments only).
export TMPDIR=/tmp
for f in build/ablate/lib_*.so; do
  n=$(basename $f .so)
  PGAS_HIP_LIB=$PWD/$f timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/abl_$n -- python3 bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-profile --T 40 > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/abl_$n/*/*counter_collection.csv")[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Kernel_Name'].startswith('void k_fused'):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print("$n", {k: round(sum(v)/len(v)/4096,1) for k,v in acc.items()})
PY
done
