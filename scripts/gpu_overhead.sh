#!/bin/bash
for n in 100000 1250000 2500000 5000000; do
  python bench.py --points $n --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['config']['points'], 'ms/step', j['ms_per_step'], 'nn', j['roofline']['kernel_ms_avg'], 'reduce', j['roofline']['reduce_ms_avg'], 'it/s', j['value'])"
done
MI_ICP_FORCE_COMM=1 python bench.py --points 1250000 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | grep "metric" | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rccl1', j['config']['points'], 'ms/step', j['ms_per_step'])"
