#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in plain symbolic; do
  if [ $v = symbolic ]; then export SYMBOLIC=1; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/al_$v -- python3 tools/alg1_launches.py 201 > /dev/null 2>&1
  python3 -c "
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/al_$v/*/*kernel_stats.csv')[0])))
tot=sum(int(r['Calls']) for r in rows)
print('$v', 'launches per step: %.1f' % (tot/200.0))
for r in sorted(rows, key=lambda r:-int(r['Calls']))[:40]:
    print('  %6.2f  %s' % (int(r['Calls'])/200.0, r['Name'][:110]))" > gpurun_out/alg1_launches_$v.txt
  rm -rf gpurun_out/al_$v
  head -45 gpurun_out/alg1_launches_$v.txt
done
