#!/bin/bash
# Kernel timeline of the last RegistrationICP call of measure_latency.py (env passes through).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${LAT_N:-10000000}
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_call -o c -- python $R/scripts/measure_latency.py $N > $R/gpurun_out/prof_call.log 2>&1
cd $R
grep '^{' gpurun_out/prof_call.log | cut -c60-260
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_call/c_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'].split('(')[0][:50] for r in rows]
idx=[i for i,n in enumerate(names) if 'kd_build_groups' in n][-1]
t0=int(rows[idx]['Start_Timestamp'])
skip=('rs_histogram','scan_','rs_scatter')
for r,n in list(zip(rows,names))[idx:idx+60]:
    if any(s in n for s in skip): continue
    print('%9.1f %9.1f  %s  grid=%s' % ((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,n, r.get('Grid_Size_X','')))
PY
