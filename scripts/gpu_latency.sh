#!/bin/bash
# One RegistrationICP call on small clouds: rows (measure_latency.py) + where the time goes
# (rocprofv3 kernel trace of the last call: kernel time vs gaps).  LAT_N = points.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${LAT_N:-307200}
cd $R; mkdir -p gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lat -o l -- python $R/scripts/measure_latency.py $N > $R/gpurun_out/prof_lat.log 2>&1
cd $R
grep '^{' gpurun_out/prof_lat.log | cut -c1-220
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_lat/l_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
last_build=max(i for i,n in enumerate(names) if 'kd_build_groups' in n)
b0=last_build
while b0>0 and 'reduce' not in names[b0-1] and 'nn_packet' not in names[b0-1]: b0-=1
seq=rows[b0:]
def span(s): return (int(s[-1]['End_Timestamp'])-int(s[0]['Start_Timestamp']))/1e3
def ksum(s): return sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in s)/1e3
print('last call: %d kernels, span %.1f us, kernel time %.1f us' % (len(seq), span(seq), ksum(seq)))
nn=[i for i,r in enumerate(seq) if 'nn_packet_kernel<true' in r['Kernel_Name'] or 'icp_small_iteration' in r['Kernel_Name']]
it=seq[nn[1]:]           # from the second seeded pass on: the steady iterations
n_it=sum(1 for r in it if 'nn_packet_kernel<true' in r['Kernel_Name'] or 'icp_small_iteration' in r['Kernel_Name'])
print('steady iterations: %d, span %.1f us (%.1f per iteration), kernel time %.1f us (%.1f per iteration)' % (n_it, span(it), span(it)/n_it, ksum(it), ksum(it)/n_it))
t0=int(seq[0]['Start_Timestamp'])
for r in seq[:70]:
    print('%9.1f %8.1f  %s' % ((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Kernel_Name'].split('(')[0][:50]))
PY
