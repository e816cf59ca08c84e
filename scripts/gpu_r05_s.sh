#!/bin/bash
# Round 5, run s: the odometry step's solve by a wave against one thread (temporary switch), same box; all GPU tests.
O=gpurun_out/r05s
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 > $O/tests_full.log 2>&1; grep -E "passed|failed|error" $O/tests_full.log | tail -3 | tee $O/tests.txt
{
for v in wave serial wave serial; do
  if [ $v = serial ]; then export MI_ICP_OD_SERIAL_SOLVE=1; else unset MI_ICP_OD_SERIAL_SOLVE; fi
  echo "== $v"; timeout 120 python scripts/measure_odometry.py 2>&1 | grep -v amdgpu.ids | cut -c1-130
done
} | tee $O/od_solve_ab.txt
