#!/bin/bash
# colored-ICP row: parity tests + the C++ surface test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_colored.py tests/test_gpu_cpp.py -x -q 2>&1 | tail -40 > gpurun_out/colored_tests.log
cat gpurun_out/colored_tests.log
