#!/bin/bash
# same-box A/B of two builds on the rows the tree walk dominates: normals, k-NN, first pass, cold call, transient, noisy
O=gpurun_out/ab_touch; mkdir -p $O
one() {
  python scripts/measure_normals_10m.py 2>&1 | grep normals
  python scripts/measure_knn.py 30,0.0 100,0.0 2>&1 | grep '^{' | cut -c50-140
  python bench.py --no-cpu-baseline 2>&1 | grep '^{"metric' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['config']['secondary']
print('head', j['value'], 'nn', j['roofline']['kernel_ms_avg'], 'cold', s['cold_30_iteration_call_ms'], 'first', s['first_pass_ms'], 'first+halos', s['first_pass_with_halos_ms'], 'noisy.15 nn', s['noisy_sigma_0.15_nn_ms'], 'transient', s['transient_30_iteration_loop_ms'])"
  python scripts/measure_noisy.py 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('sigma', j['sigma_over_spacing'], 'nn %.4f it %.4f' % (j['nn_ms'], j['ms_per_iter']))"
  python scripts/dev/halo_build_time.py 2>&1 | tail -2
}
for which in ${AB_LIBS:-base new base new}; do
  echo "== $which"
  MI_ICP_LIB_PATH=$PWD/cupoch_amd/lib/ab_$which.so one
done 2>&1 | tee $O/ab.txt
