#!/bin/bash
# A/B of the first pass: own seeds (greedy descent + seeded search) vs a plain unseeded pass.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seeded.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -3
for N in 100000 307200 1000000 10000000; do
echo "== default n=$N"; python scripts/measure_latency.py $N | cut -c60-260
echo "== no coarse first"; MI_ICP_NO_COARSE_FIRST=1 python scripts/measure_latency.py $N | cut -c60-260
done
LAT_N=10000000 bash scripts/gpu_trace_call.sh
