#!/bin/bash
# reduction sweeping every XCD's eighth downwards (MI_ICP_REDUCE_ORDER=1) against the ascending grid-stride sweep
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do for o in 0 1; do
  MI_ICP_REDUCE_ORDER=$o timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('order $o: %.1f it/s  ms %.4f  nn %.4f  reduce %.4f  traffic %s' % (d['value'], d['ms_per_step'], r['kernel_ms_avg'], r['reduce_ms_avg'], r.get('traffic')))"
done; done | tee gpurun_out/j_order.txt
MI_ICP_REDUCE_ORDER=1 timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/j_tests.log
for n in 1000000 100000000; do for o in 0 1; do
  MI_ICP_REDUCE_ORDER=$o timeout 300 python bench.py --points $n --steps 50 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('n=$n order $o: %.1f it/s  ms %.4f  nn %.4f  reduce %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms_avg'], r['reduce_ms_avg']))"
done; done | tee -a gpurun_out/j_order.txt
