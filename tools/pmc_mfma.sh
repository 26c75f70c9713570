#!/bin/bash
# Instruction mix of the M = 729 k_propagate, vector form against matrix-core form:
#   tools/pmc_mfma.sh TAG [ENV=VAL ...]  -> gpurun_out/pmc_mfma_TAG.txt     (TAG valu: PGAS_MFMA_PROPAGATE=0, TAG mfma: =1)
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $R
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_$TAG -- python3 bench.py --workload emps --steps 1 --warmup 0 --cpu-steps 0 --no-profile --T 100 > /dev/null 2> gpurun_out/pmc_mfma_$TAG.err || { tail -5 gpurun_out/pmc_mfma_$TAG.err; exit 1; }
python3 tools/pmc_kernels.py gpurun_out/pmc_mfma_$TAG raw | grep "k_step\|k_propagate" > gpurun_out/pmc_mfma_$TAG.txt
rm -rf gpurun_out/pmc_mfma_$TAG
cat gpurun_out/pmc_mfma_$TAG.txt
