#!/bin/bash
# Same-box A/B (old = ab_pow2.so: commit e23b0b1, 2^d cells at <= 2/3 fill) of what is NOT the headline: noisy loop, transient, cold call
O=gpurun_out/r05i; mkdir -p $O
for rep in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export MI_ICP_LIB_PATH=$PWD/cupoch_amd/lib/ab_pow2.so; else unset MI_ICP_LIB_PATH; fi
  echo "== $lib (repetition $rep)"
  timeout 300 python scripts/measure_noisy.py 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  sigma %.2f: search %.4f ms, iteration %.4f ms' % (d['sigma_over_spacing'], d['nn_ms'], d['ms_per_iter']))"
  timeout 300 python scripts/dev/transient_one.py 2>/dev/null | tail -2
  timeout 300 python scripts/measure_latency.py 10000000 2>/dev/null | grep '^{' | cut -c1-400
done; done 2>&1 | tee $O/ab_layouts_other.txt
