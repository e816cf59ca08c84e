#!/bin/bash
# Same-box A/B: the library before the 80 %-fill layouts (cupoch_amd/lib/ab_pow2.so = commit e23b0b1: 2^d cells at <= 2/3 fill,
# sampled medians) against the shipped one, alternating, at 10M / 20M / 100M points
O=gpurun_out/r05g; mkdir -p $O
for rep in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export MI_ICP_LIB_PATH=$PWD/cupoch_amd/lib/ab_pow2.so; else unset MI_ICP_LIB_PATH; fi
  for n in 10000000 20000000 100000000; do
    timeout 600 python bench.py --points $n --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric' | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('$lib rep $rep n=%d: %.1f it/s, %.4f ms/step, search %.4f, reduce %.4f, build %.1f ms' % (j['config']['points'], j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['reduce_ms_avg'], j['config']['build_ms']))"
  done
done; done 2>&1 | tee $O/ab_layouts.txt
