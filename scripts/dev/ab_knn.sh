#!/bin/bash
# same-box A/B of builds on the k-NN rows
for which in ${AB_LIBS:-base new base new}; do
  echo "== $which"
  export MI_ICP_LIB_PATH=$PWD/cupoch_amd/lib/ab_$which.so
  python scripts/measure_normals_10m.py 2>&1 | grep normals
  python scripts/measure_knn.py 30,0.0 64,0.0 100,0.0 2>&1 | grep '^{' | cut -c50-140
done 2>&1 | tee gpurun_out/ab_knn.txt
