#!/bin/bash
# RGB-D odometry: per-kernel stats (rocprofv3) of scripts/measure_odometry.py
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_odo -o o -- python $R/scripts/measure_odometry.py > $R/gpurun_out/odo_prof.log 2>&1
echo "rc=$?"; head -14 $R/gpurun_out/prof_odo/o_kernel_stats.csv | cut -c1-150
