#!/bin/bash
# KinFu-side callers: measured rows + per-kernel stats (rocprofv3) of the same script
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/scripts/measure_kinfu.py 2>/dev/null | grep "^{" > $R/gpurun_out/kinfu_measured.jsonl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kinfu -o k -- python $R/scripts/measure_kinfu.py > $R/gpurun_out/kinfu_prof.log 2>&1
echo "rc=$?"; ls $R/gpurun_out/prof_kinfu; head -12 $R/gpurun_out/prof_kinfu/k_kernel_stats.csv | cut -c1-200
