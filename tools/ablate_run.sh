#!/bin/bash
# Times the sweep with each ablated library build (build/ablate/lib_*.so); results are NOT valid, only the timing is.
for f in build/ablate/lib_*.so; do
  echo "== $f"
  PGAS_HIP_LIB=$PWD/$f timeout -k 10 120 python bench.py --steps 2 --warmup 1 --cpu-steps 0 --T ${ABL_T:-500} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('ms_per_sweep %.2f  fused_us %.2f' % (d['ms_per_step'], d['roofline']['avg_launch_us']))"
done
