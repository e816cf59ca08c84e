#!/bin/bash
# Same-box A/B of two builds of libmi_icp.so (cupoch_amd/lib/ab_head.so, ab_new.so; the box copy is scratch):
# boxes differ by +-2 %, so builds are compared alternating on one box.  AB_CMD = what to run per build,
# AB_LIBS = which builds (ab_<name>.so).
O=gpurun_out/ab_libs
mkdir -p $O
AB_CMD=${AB_CMD:-"python scripts/measure_latency.py 20000 1000000 2>&1 | grep '^{' | cut -c60-115,200-260; python bench.py --no-cpu-baseline --no-secondary 2>&1 | grep '^{\"metric' | python scripts/benchline.py"}
cp cupoch_amd/_lib.py /tmp/_lib_new.py
for round in 1 2; do
for which in ${AB_LIBS:-head new}; do
  cp cupoch_amd/lib/ab_$which.so cupoch_amd/lib/libmi_icp.so
  # (a symbol the older build lacks must not stop its load)
  if [ $which = head ]; then sed '/mi_icp_debug_solve_both/d' /tmp/_lib_new.py > cupoch_amd/_lib.py; else cp /tmp/_lib_new.py cupoch_amd/_lib.py; fi
  touch cupoch_amd/lib/libmi_icp.so
  echo "== $which ($round)"
  timeout 600 bash -c "$AB_CMD"
done
done 2>&1 | tee $O/ab.txt
