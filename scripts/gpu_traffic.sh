#!/bin/bash
# HBM traffic of the search kernel on the bench workload, with the FETCH_SIZE / WRITE_SIZE calibration
# MI355X_MICROARCH.md prescribes (known byte counts in the kernel's own access shapes).
#   -> gpurun_out/fetch_calibration.txt, gpurun_out/nn_traffic.json   (copied into profiles/ by hand)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 $R/scripts/ubench/fetch_calib.hip -o /tmp/fetch_calib || exit 1
cd /tmp
/tmp/fetch_calib > $R/gpurun_out/calib_known.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/calib_$c -o c -- /tmp/fetch_calib > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/nnpmc_$c -o p -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/nnpmc_$c.log 2>&1
done
cd $R
python scripts/traffic_summary.py | tee gpurun_out/fetch_calibration.txt
