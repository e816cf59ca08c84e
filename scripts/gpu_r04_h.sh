#!/bin/bash
# first halo ring in 64 bytes: invariants, parity subset, noisy search times, transient
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_tree_invariants.py tests/test_gpu_seeded.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/h_tests.log
timeout 300 python scripts/measure_noisy.py > gpurun_out/h_noisy.log 2>&1
timeout 300 python scripts/dev/transient_census.py > gpurun_out/h_transient.log 2>&1
timeout 300 python bench.py --steps 100 --warmup 20 > gpurun_out/h_bench.log 2>&1
tail -3 gpurun_out/h_tests.log; tail -8 gpurun_out/h_noisy.log; tail -5 gpurun_out/h_transient.log; tail -1 gpurun_out/h_bench.log
