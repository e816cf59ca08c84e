#!/bin/bash
# leaf regions moved into the leaf lines' fourth row (original indices into an array of their own): whole GPU suite + headline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -8 | tee gpurun_out/l_tests.log
for k in 1 2; do
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('100 steps: %.1f it/s  ms %.4f  nn %.4f  reduce %.4f  frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_avg'], r['reduce_ms_avg'], r['frac']))"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('20 steps: %.1f it/s  ms %.4f  nn %.4f  reduce %.4f  frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_avg'], r['reduce_ms_avg'], r['frac']))"
done | tee gpurun_out/l_bench.txt
timeout 300 python scripts/measure_noisy.py 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('noisy', d['sigma_over_spacing'], round(d['nn_ms'],4), round(d['ms_per_iter'],4))" | tee -a gpurun_out/l_bench.txt
