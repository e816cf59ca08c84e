#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r01 -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/rocprof_stats.log 2>&1
cd $R
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof_stats/r01_kernel_stats.csv')):
    print(r['Name'][:58].ljust(60), r['Calls'].rjust(4), ('%.1f us avg' % (float(r['AverageNs'])/1e3)).rjust(14), ('%.2f ms tot' % (float(r['TotalDurationNs'])/1e6)).rjust(12))
PY
