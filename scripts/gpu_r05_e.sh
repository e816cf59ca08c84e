#!/bin/bash
O=gpurun_out/r05e
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests/test_gpu_tree_invariants.py tests/test_gpu_robustness.py -m gpu -q --timeout=600 > $O/t_tree.log 2>&1; echo "tree tests rc=$?"; grep -E "passed|failed|^FAILED|assert" $O/t_tree.log | tail -8
timeout 600 python bench.py --no-cpu-baseline --big-points 0 2> $O/bench.err | grep '^{"metric' | tee $O/bench.json | python scripts/benchline.py
python -c "
import json;d=json.load(open('$O/bench.json'))['config']['secondary'];print({k:v for k,v in d.items() if not isinstance(v,(dict,str)) or 'kind' in k})"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_cold -o s -- python $GRAFT_REPO_ROOT/scripts/measure_latency.py 10000000 > $GRAFT_REPO_ROOT/$O/st_cold.log 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05e/st_cold/s_kernel_trace.csv')))
seq=[]
for r in rows:
    n=r['Kernel_Name']
    if n.startswith('mi::hp_') or 'kd_build_groups' in n or 'cells_' in n or 'rs_' in n:
        seq.append((int(r['Start_Timestamp']), n.split('(')[0], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
seq.sort()
last=[i for i,s in enumerate(seq) if 'cells_sample_gather' in s[1]][-1]
tot=0
for s in seq[last:last+48]:
    print('%-28s %8.1f us' % (s[1][:28], s[2])); tot+=s[2]
    if 'kd_build_groups' in s[1]: break
print('sum', tot)
PY
find $O -name "*.db" -delete 2>/dev/null
