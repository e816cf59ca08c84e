#!/bin/bash
# The wave-wide step (wave_solver.h): parity subset + what it changes in call latency and the headline.
O=gpurun_out/wave_step
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wave_solver.py tests/test_gpu_parity.py tests/test_gpu_seeded.py tests/test_gpu_distributed.py tests/test_gpu_python_api.py tests/test_gpu_depth.py -m gpu -q -x --timeout=600 > $O/t.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/t.log
timeout 300 python scripts/measure_latency.py 20000 100000 1000000 2>&1 | grep '^{' | tee $O/latency.jsonl | cut -c60-300
timeout 600 python bench.py --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric' | tee $O/bench.json | python scripts/benchline.py
