#!/bin/bash
# Cold RegistrationICP call on 10M-point clouds: rows + per-kernel time of the whole run.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${LAT_N:-10000000}
cd $R; mkdir -p gpurun_out
if [ -n "$COLD_TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_tree_invariants.py tests/test_gpu_seeded.py -x -q 2>&1 | tail -3; fi
python scripts/measure_latency.py $N | cut -c1-260
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cold -o c -- python $R/scripts/measure_latency.py $N > $R/gpurun_out/prof_cold.log 2>&1
cd $R
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_cold/c_kernel_stats.csv')))
for r in rows[:22]:
    print(r['Name'].split('(')[0][:60].ljust(62), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs'])/1e3)).rjust(10), 'us avg', ('%.1f' % (float(r['MinNs'])/1e3)).rjust(9), 'min')
PY
