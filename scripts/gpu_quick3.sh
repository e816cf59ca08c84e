#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -x > gpurun_out/t_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/t_gpu.log
timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_10m.log
bash scripts/gpu_small.sh 2>&1 | tail -22
