#!/bin/bash
# HBM traffic of the dense VoxelDownSample's kernels at 10M points (rocprofv3 --pmc, one counter per pass as the guide prescribes);
# -> gpurun_out/vxpmc/summary.txt (copied to profiles/r06_voxel_dense_traffic.txt)
R=$GRAFT_REPO_ROOT; O=gpurun_out/vxpmc; mkdir -p $R/$O; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/$c -o p -- python $R/scripts/dev/voxel_one.py > $R/$O/$c.log 2>&1); echo "pmc $c rc=$?"
done
cd $R
{ echo "# FETCH_SIZE / WRITE_SIZE (KB per launch as reported; vector loads count at half their bytes on gfx950: profiles/r06_fetch_calibration.txt) of scripts/dev/voxel_one.py: 10M points, voxel 0.01, six calls points only, six with normals"
  python scripts/pmc_kernels.py "$O/*SIZE/p_counter_collection.csv" vx_ bounds_partial; } | tee $O/summary.txt
find $O -name "*.db" -delete 2>/dev/null
