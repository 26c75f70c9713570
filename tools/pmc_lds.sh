#!/bin/bash
# LDS bank-conflict counters of the sweep's kernels: tools/pmc_lds.sh TAG [ENV=VAL ...]  -> gpurun_out/pmc_lds_TAG.txt
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $R
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_lds_$TAG -- python3 bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-profile --T 200 > /dev/null 2> gpurun_out/pmc_lds_$TAG.err || { tail -5 gpurun_out/pmc_lds_$TAG.err; exit 1; }
python3 tools/pmc_kernels.py gpurun_out/pmc_lds_$TAG raw | grep "k_step\|k_groups\|k_propagate" > gpurun_out/pmc_lds_$TAG.txt
rm -rf gpurun_out/pmc_lds_$TAG
cat gpurun_out/pmc_lds_$TAG.txt
