#!/bin/bash
# quick GPU loop: parity + census + 10M bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_0_primitives.py tests/test_gpu_parity.py -m gpu -q --timeout=400 > gpurun_out/t_all.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/t_all.log
timeout 300 python scripts/nn_census.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/census.log
timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_10m.log
