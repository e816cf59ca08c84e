#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/t_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -6 gpurun_out/t_gpu.log
timeout 300 python scripts/nn_census.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/census.log
timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_10m.log
echo "--- MI_ICP_NO_RESORT=1"
MI_ICP_NO_RESORT=1 timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-resort', 'ms/step', j['ms_per_step'], 'nn', j['roofline']['kernel_ms_avg'])"
echo "--- MI_ICP_NO_KD=1"
MI_ICP_NO_KD=1 timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-kd', 'ms/step', j['ms_per_step'], 'nn', j['roofline']['kernel_ms_avg'], 'build', j['config']['build_ms'])"
