#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 300 python scripts/measure_latency.py > gpurun_out/latency.jsonl 2> gpurun_out/latency.err
cat gpurun_out/latency.jsonl; tail -3 gpurun_out/latency.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lat -o l -- python $R/scripts/measure_latency.py 307200 > $R/gpurun_out/prof_lat.log 2>&1
cd $R
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_lat/l_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last call: find last build_leaves occurrence (set_target of the final repetition)
idx=[i for i,r in enumerate(rows) if 'bounds' in r['Kernel_Name'].lower()]
names=[r['Kernel_Name'] for r in rows]
# locate the start of the last set_target: search backwards for the 2nd-from-last morton sequence
starts=[i for i,r in enumerate(rows) if 'build_leaves' in r['Kernel_Name']]
s=starts[-1]
# go back to the first kernel of that set_target (bounds kernel before it)
b=max(i for i in idx if i<s and (s-i)<80)
b0=b
while b0-1 in idx: b0-=1
seq=rows[b0:]
t0=int(seq[0]['Start_Timestamp'])
prev=None; tot_k=0; 
print('kernels in last call:',len(seq))
agg={}
for r in seq:
    st,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    k=r['Kernel_Name'].split('(')[0][:40]
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=(e-st)/1e3
    tot_k+=(e-st)/1e3
print('span %.1f us, sum of kernel durations %.1f us' % ((int(seq[-1]['End_Timestamp'])-t0)/1e3, tot_k))
for k,(c,d) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]:
    print(k.ljust(42), str(c).rjust(4), ('%.1f us' % d).rjust(12))
PY
