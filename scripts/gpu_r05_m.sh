#!/bin/bash
# Round 5, run m: what the host's look at the loop every 8 iterations costs -- the shard emulation and the headline with chunks of 8 / 16 / 64.
O=gpurun_out/r05m
mkdir -p $O
export TMPDIR=/tmp
for ch in 8 64 16 8 64; do
  echo "== MI_ICP_CHUNK=$ch"
  MI_ICP_CHUNK=$ch timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('  ranks %d: step %.4f nn %.4f reduce %.4f' % (r['ranks'], r['ms_per_step_compute_only'], r['nn_ms'], r['reduce_ms']))"
  MI_ICP_CHUNK=$ch timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>&1 | grep '^{"metric' | python scripts/benchline.py
done 2>&1 | tee $O/chunk.txt
