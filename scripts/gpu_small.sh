#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small -o s -- python $R/bench.py --points 1250000 --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_small.log 2>&1
cd $R
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof_small/s_kernel_stats.csv')):
    if float(r['Percentage'])>0.5: print(r['Name'][:50].ljust(52), r['Calls'].rjust(4), ('%.1f us' % (float(r['AverageNs'])/1e3)).rjust(10), r['Percentage'])
rows=list(csv.DictReader(open('gpurun_out/prof_small/s_kernel_trace.csv')))
rows=[r for r in rows if 'nn_packet' in r['Kernel_Name'] or 'reduce_kernel' in r['Kernel_Name'] or 'loop_step' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None
out=[]
for r in rows[60:75]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap = (s-prev)/1e3 if prev else 0
    out.append('%s dur %.1f us gap %.1f us' % (r['Kernel_Name'][:28], (e-s)/1e3, gap)); prev=e
print('\n'.join(out))
PY
