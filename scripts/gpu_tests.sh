#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 "$@" > gpurun_out/t_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -15 gpurun_out/t_gpu.log
