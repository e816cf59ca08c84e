#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_cpp.py -m gpu -q --timeout=400 2>&1 | tail -15 | tee gpurun_out/t_cpp.log
timeout 900 python scripts/measure_configs.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/configs.log
# RCCL path with a single rank (the communicator + all-reduce are exercised; 8-GPU runs are the driver's)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --points 2000000 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/bench_dist1.log
