#!/bin/bash
# Round 5: TRI cell layouts + histogram planes -- tests, build times, bench
O=gpurun_out/r05d
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests/test_gpu_tree_invariants.py tests/test_gpu_robustness.py tests/test_gpu_0_primitives.py -m gpu -q -x --timeout=600 > $O/t_tree.log 2>&1; echo "tree tests rc=$?"; grep -E "passed|failed|Error|assert" $O/t_tree.log | tail -8
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/t_gpu.log | tail -12
timeout 600 python bench.py --no-cpu-baseline --big-points 0 2> $O/bench.err | grep '^{"metric' | tee $O/bench.json | python scripts/benchline.py
python -c "
import json;d=json.load(open('$O/bench.json'))['config']['secondary'];print({k:v for k,v in d.items() if not isinstance(v,(dict,str)) or 'kind' in k})"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_cold -o s -- python $GRAFT_REPO_ROOT/scripts/measure_latency.py 10000000 > $GRAFT_REPO_ROOT/$O/st_cold.log 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r05d/st_cold/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows:
        n=r['Name']
        if any(k in n for k in ('kd_build','hp_','cells_','rs_scatter','rs_hist','build_level','tree_scale','gather_source','bounds')):
            print('%-70s calls %4s avg %9.1f us' % (n[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
find $O -name "*.db" -delete 2>/dev/null
du -sh $O | tail -1
