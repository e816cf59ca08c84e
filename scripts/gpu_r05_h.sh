#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
for lib in old new; do
  if [ $lib = old ]; then export MI_ICP_LIB_PATH=$PWD/cupoch_amd/lib/ab_pow2.so; else unset MI_ICP_LIB_PATH; fi
  for links in default nolinks; do
    if [ $links = nolinks ]; then export MI_ICP_NO_LINKS=1; else unset MI_ICP_NO_LINKS; fi
    timeout 600 python bench.py --points 100000000 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric' | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('$lib $links: %.1f it/s, %.4f ms/step (min %.4f max %.4f), search %.4f, reduce %.4f' % (j['value'], j['ms_per_step'], j['ms_per_step_min_max'][0], j['ms_per_step_min_max'][1], j['roofline']['kernel_ms_avg'], j['roofline']['reduce_ms_avg']))"
  done
done 2>&1 | tee $O/ab_100m.txt
