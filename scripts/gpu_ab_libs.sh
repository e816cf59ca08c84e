#!/bin/bash
# Same-box A/B of builds of libmi_icp.so (cupoch_amd/lib/ab_<name>.so): boxes differ by +-2 %, so builds are
# compared alternating on one box.  AB_CMD = what to run per build, AB_LIBS = which builds.
O=gpurun_out/ab_libs
mkdir -p $O
AB_CMD=${AB_CMD:-"python scripts/measure_latency.py 20000 1000000 2>&1 | grep '^{' | cut -c60-115,200-260; python bench.py --no-cpu-baseline --no-secondary 2>&1 | grep '^{\"metric' | python scripts/benchline.py"}
for round in 1 2; do
for which in ${AB_LIBS:-head new}; do
  echo "== $which ($round)"
  MI_ICP_LIB_PATH=$PWD/cupoch_amd/lib/ab_$which.so timeout 600 bash -c "$AB_CMD"
done
done 2>&1 | tee $O/ab.txt
