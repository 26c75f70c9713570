#!/bin/bash
# SQ instruction / cycle counters of the sweep's kernels (rocprofv3 --pmc, separate passes, program directly after --).
# usage (on the GPU box): tools/pmc_sq.sh TAG [bench args...]   -> gpurun_out/TAG/pmc_sq_*.txt
set -o pipefail
TAG=${1:-r02}; shift
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --cpu-steps 0 --no-profile --T 200 $@"
[ -f $O/counters_available.txt ] || rocprofv3 -L > $O/counters_available.txt 2>&1
pass() {  # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -- python3 bench.py $ARGS > /dev/null 2> $O/pmc_$name.err || { tail -5 $O/pmc_$name.err; return 1; }
  python3 tools/pmc_kernels.py $O/pmc_$name raw > $O/pmc_sq_$name.txt && cat $O/pmc_sq_$name.txt
  rm -rf $O/pmc_$name
}
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F64 || pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
pass cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU || pass cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 || true
