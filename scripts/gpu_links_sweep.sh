#!/bin/bash
# neighbour-list start bound (MI_ICP_LINK_DELTA, a fraction of the leaf-level node's extent): build cost vs search cost
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 0.2 0.25 0.3; do
  echo "== MI_ICP_LINK_DELTA=$d"
  cd /tmp
  MI_ICP_LINK_DELTA=$d rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sweep_$d -o s -- python $R/scripts/measure_noisy.py > $R/gpurun_out/sweep_$d.log 2>&1
  cd $R
  grep -E "leaf_links" gpurun_out/sweep_$d/s_kernel_stats.csv | sed "s/(.*)\"//" | cut -c1-100
  grep sigma gpurun_out/sweep_$d.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['sigma_over_spacing'], 'nn_ms %.4f' % j['nn_ms'], 'rec/pkt %.2f' % j['records_per_packet'], 'batches %.2f' % j['leaf_batches_per_packet'])"
done
