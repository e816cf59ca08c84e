#!/bin/bash
for which in ${AB_LIBS:-cur}; do
  echo "== $which"
  MI_ICP_LIB_PATH=$PWD/cupoch_amd/lib/ab_$which.so python scripts/measure_normals_10m.py 2>&1 | grep normals
done
