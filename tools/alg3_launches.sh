#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/al3 -- python3 tools/alg3_launches.py 201 > /dev/null 2>&1
python3 -c "
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/al3/*/*kernel_stats.csv')[0])))
tot=sum(int(r['Calls']) for r in rows)
print('Algorithm3 launches per step (incl. the per-trajectory statistics of Algorithm2): %.1f' % (tot/200.0))
for r in sorted(rows, key=lambda r:-int(r['Calls']))[:45]:
    print('  %6.2f  %s' % (int(r['Calls'])/200.0, r['Name'][:110]))" > gpurun_out/alg3_launches.txt
rm -rf gpurun_out/al3
head -50 gpurun_out/alg3_launches.txt
