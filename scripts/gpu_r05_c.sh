#!/bin/bash
# Round 5: the halo with 4 stored lines against the shipped 8, same box, alternating
O=gpurun_out/r05c
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
python -c "
from cupoch_amd import _lib
print(_lib.build(variant='h4', defines=('-DMI_HALO_STORED=4',)))" >> $O/build.log 2>&1; echo "variant rc=$?"
for rep in 1 2; do
for lib in 8 4; do
  if [ $lib = 4 ]; then export MI_ICP_LIB_PATH=$PWD/cupoch_amd/lib/libmi_icp_h4.so; else unset MI_ICP_LIB_PATH; fi
  echo "== stored lines: $lib (repetition $rep)"
  timeout 300 python scripts/dev/halo_build_time.py 2>/dev/null | tail -2
  timeout 300 python scripts/measure_noisy.py 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  sigma %.2f: search %.4f ms, iteration %.4f ms, halo lines/packet %.1f, packets that walk %.2f %%' % (d['sigma_over_spacing'], d['nn_ms'], d['ms_per_iter'], d['halo_lines_per_packet'], 100*d['packets_that_walk']))"
  timeout 300 python scripts/dev/transient_one.py 2>/dev/null | tail -2
done; done 2>&1 | tee $O/halo_lines_4_vs_8.txt
unset MI_ICP_LIB_PATH
du -sh $O | tail -1
