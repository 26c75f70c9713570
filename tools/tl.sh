#!/bin/bash
# kernel-trace timeline of one bench run: tools/tl.sh NAME [ENV=VAL ...]   -> gpurun_out/tl_NAME.txt
name=$1; shift
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl_$name
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$name -o tr -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-profile > $R/gpurun_out/tl_$name.json 2> $R/gpurun_out/tl_$name.err
python3 $R/tools/timeline.py $R/gpurun_out/tl_$name > $R/gpurun_out/tl_$name.txt 2>&1
rm -rf $R/gpurun_out/tl_$name
cat $R/gpurun_out/tl_$name.txt
