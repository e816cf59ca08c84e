#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=900 -x 2>&1 | tail -2
for n in 1250000 10000000; do
MI_ICP_BENCH_NO_EVENTS=1 python bench.py --points $n --steps 48 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('noevents', j['config']['points'], 'ms/step', j['ms_per_step'], 'nn', j['roofline']['kernel_ms_avg'], 'reduce', j['roofline']['reduce_ms_avg'], 'it/s', j['value'])"
done
MI_ICP_FORCE_COMM=1 MI_ICP_BENCH_NO_EVENTS=1 python bench.py --points 1250000 --steps 48 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('noevents rccl1', j['config']['points'], 'ms/step', j['ms_per_step'])"
cd /tmp
MI_ICP_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small -o s -- python $R/bench.py --points 1250000 --steps 48 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_small.log 2>&1
cd $R
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_small/s_kernel_trace.csv')))
rows=[r for r in rows if 'nn_packet' in r['Kernel_Name'] or 'reduce_kernel' in r['Kernel_Name'] or 'loop_step' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None
for r in rows[60:72]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%s dur %.1f us gap %.1f us' % (r['Kernel_Name'][:28], (e-s)/1e3, (s-prev)/1e3 if prev else 0)); prev=e
PY
