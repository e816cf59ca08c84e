#!/bin/bash
O=gpurun_out/r05j; mkdir -p $O
run() { timeout 600 python scripts/measure_shard.py 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ranks %d: step %.4f nn %.4f reduce %.4f' % (d['ranks'], d['ms_per_step_compute_only'], d['nn_ms'], d['reduce_ms']))"; }
for rep in 1 2; do
echo "== old library (e23b0b1)"; MI_ICP_LIB_PATH=$PWD/cupoch_amd/lib/ab_pow2.so run
echo "== new, default layout (TRI at 10M)"; run
echo "== new, MI_ICP_CELL_LAYOUT=pow2 (4096 cells... 80 % rule: 4096 at 10M)"; MI_ICP_CELL_LAYOUT=pow2 run
echo "== new, MI_ICP_CELL_LAYOUT=r4 (2/3 fill)"; MI_ICP_CELL_LAYOUT=r4 run
done 2>&1 | tee $O/shard_ab.txt
