#!/bin/bash
# Round 6: one rank's share by shard size, the library before (2 in flight at every size, the counter copied every chunk) and
# after (4 in flight below 4M points, the counter every eighth chunk of a loop that has declined), alternating on one box
O=gpurun_out/r06h
mkdir -p $O
for i in 1 2; do
  echo "== after"; timeout 300 python scripts/measure_shard.py 2>&1 | grep '^{' | tee $O/after_$i.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ranks', d['ranks'], 'step', d['ms_per_step_compute_only'], 'nn', d['nn_ms'], 'red', d['reduce_ms'], 'x', d['speedup_vs_1_rank'])"
  echo "== before"; MI_ICP_LIB_PATH=cupoch_amd/lib/ab_prev.so timeout 300 python scripts/measure_shard.py 2>&1 | grep '^{' | tee $O/before_$i.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ranks', d['ranks'], 'step', d['ms_per_step_compute_only'], 'nn', d['nn_ms'], 'red', d['reduce_ms'], 'x', d['speedup_vs_1_rank'])"
done
