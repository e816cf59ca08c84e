#!/bin/bash
# Round 6: several iterations per launch for small clouds (fused_small.h icp_small_loop_kernel): parity, then call latencies
# and the KinFu step with and without (MI_ICP_NO_SMALL_LOOP=1), same box
O=gpurun_out/r06f
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_seeded.py -x -q --timeout=300 -k "several_iterations or one_launch" 2>&1 | tail -3
for i in 1 2; do
  echo "== loop"; timeout 300 python scripts/measure_latency.py 5000 20000 40000 60000 2>&1 | grep '^{' | tee $O/lat_loop_$i.jsonl | cut -c1-230
  echo "== launches"; MI_ICP_NO_SMALL_LOOP=1 timeout 300 python scripts/measure_latency.py 5000 20000 40000 60000 2>&1 | grep '^{' | tee $O/lat_launches_$i.jsonl | cut -c1-230
done
echo "== kinfu loop"; timeout 300 python scripts/measure_kinfu.py 2>&1 | grep '^{' | grep KinFu | cut -c1-200
echo "== kinfu launches"; MI_ICP_NO_SMALL_LOOP=1 timeout 300 python scripts/measure_kinfu.py 2>&1 | grep '^{' | grep KinFu | cut -c1-200
