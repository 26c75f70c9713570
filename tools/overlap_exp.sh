#!/bin/bash
# overlap experiments (development aid): propagate register budget x stream priority x chunk
run() { echo "== $1 | $3"; env $1 timeout -k 10 120 python bench.py --steps 2 --warmup 1 --cpu-steps 0 $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('ms_per_sweep %.2f  resample_us %.2f  propagate_us/step %.2f' % (d['ms_per_step'], r['avg_launch_us'], r['second_kernel']['avg_launch_us']/r['second_kernel']['steps_per_launch']))"; }
for lib in "A=1" "PGAS_HIP_LIB=$PWD/build/ablate/lib_W3.so" "PGAS_HIP_LIB=$PWD/build/ablate/lib_W4.so"; do
  for pr in "B=1" "PGAS_B_LOWPRIO=1"; do
    for ch in "--chunk 1" "--chunk 2" "--chunk 4"; do
      run "$lib $pr" "" "$ch"
    done
  done
done
