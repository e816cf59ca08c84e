#!/bin/bash
# Round 6: 2 against 4 elements in flight in the point-to-plane reduction, same box, alternating (the headline line)
for i in 1 2 3; do
  echo -n "2 in flight: "; timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric' | python scripts/benchline.py
  echo -n "4 in flight: "; MI_ICP_LIB_PATH=cupoch_amd/lib/libmi_icp_rsweep.so MI_ICP_AB_REDUCE_U=4 MI_ICP_AB_REDUCE_GRID=512 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric' | python scripts/benchline.py
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_scale.py tests/test_gpu_distributed.py -x -q --timeout=600 2>&1 | tail -3
