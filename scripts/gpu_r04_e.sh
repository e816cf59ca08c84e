#!/bin/bash
O=gpurun_out/r04e
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|error" $O/t_gpu.log | tail -3
timeout 600 python scripts/dev/transient_one.py 2>&1 | grep loop | tee $O/transient.txt
timeout 600 python scripts/dev/transient_census.py 2>&1 | grep -E "after" | tee $O/transient_census.txt
timeout 600 python scripts/measure_noisy.py 2>&1 | grep '^{' | tee $O/noisy.jsonl | cut -c1-330
timeout 600 python bench.py --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric' | tee $O/bench_quick.json | python scripts/benchline.py
python scripts/dev/voxel_one.py 2>&1 | grep voxel | tee $O/voxel_new.txt
