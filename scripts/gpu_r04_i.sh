#!/bin/bash
# kd_build_groups: all-padding waves leave the in-wave rounds -- invariants, parity, build time (rocprof kernel stats)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_tree_invariants.py tests/test_gpu_seeded.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/i_tests.log
timeout 300 python bench.py --steps 100 --warmup 20 > gpurun_out/i_bench.log 2>&1
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/i_prof -o i -- python scripts/dev/first_pass_time.py > gpurun_out/i_prof.log 2>&1
python - <<'PY'
import sqlite3, glob
c = sqlite3.connect(glob.glob('gpurun_out/i_prof/*.db')[0])
for r in c.execute("select name, count(*), avg(end-start) from kernels group by name order by 3 desc"):
    if any(k in r[0] for k in ('kd_build', 'cells', 'leaf_halo', 'tree_scale', 'build_level', 'bounds', 'gather_source')):
        print(r[0][:50], r[1], round(r[2] / 1000, 1))
PY
tail -3 gpurun_out/i_tests.log; python - <<'PY'
import json
d = json.loads(open('gpurun_out/i_bench.log').read().strip().splitlines()[-1])
s = d['config']['secondary']
print(d['value'], d['config']['build_ms'], {k: s[k] for k in ('cold_30_iteration_call_ms', 'build_ms_target', 'build_ms_source', 'first_pass_ms', 'transient_30_iteration_loop_ms')})
PY
