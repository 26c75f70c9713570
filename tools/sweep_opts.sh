#!/bin/bash
# Timing of the sweep for several (chunk, overlap) settings (development aid).
for opt in "--no-overlap --chunk 0" "--chunk 1" "--chunk 2" "--chunk 4" "--chunk 8" "--chunk 32"; do
  echo "== $opt"
  timeout -k 10 120 python bench.py --steps 2 --warmup 1 --cpu-steps 0 $opt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('ms_per_sweep %.2f  propagate_ms %.2f  resample_us %.2f' % (d['ms_per_step'], r['avg_launch_us']/1e3, r['second_kernel']['avg_launch_us']))"
done
