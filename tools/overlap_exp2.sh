#!/bin/bash
run() { echo "== $1"; timeout -k 10 120 python bench.py --steps 2 --warmup 1 --cpu-steps 0 $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('ms_per_sweep %.2f  resample_us %.2f  propagate_us/step %.2f' % (d['ms_per_step'], r['avg_launch_us'], r['second_kernel']['avg_launch_us']/r['second_kernel']['steps_per_launch']))"; }
for lds in 0 40960 55296 57344 65536; do for ch in 1 2 4; do run "--prop-lds $lds --chunk $ch"; done; done
