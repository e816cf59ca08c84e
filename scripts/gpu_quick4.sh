#!/bin/bash
# parity (search/registration), latency per size, 10M bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_0_primitives.py tests/test_gpu_parity.py tests/test_gpu_robustness.py -m gpu -q -x --timeout=400 > gpurun_out/t_all.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_all.log
timeout 300 python scripts/measure_latency.py 2>&1 | grep "^{" | cut -c50-250 | tee gpurun_out/latency_short.log
timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/bench_10m.log
python - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_10m.log').read())
print('bench', j['value'], 'ms/step', j['ms_per_step'], 'nn', j['roofline']['kernel_ms_avg'], 'reduce', j['roofline']['reduce_ms_avg'], 'build', j['config']['build_ms'])
PY
