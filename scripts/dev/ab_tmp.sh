cd $GRAFT_REPO_ROOT
for hb in 0 9 8; do MI_ICP_VOXEL_HB=$hb python scripts/dev/voxel_one.py 1000000 0.02 2>&1 | grep voxel | sed "s/^/1M hb $hb: /"; done
for n in 3000000 30000000; do for hb in 0 10 11; do MI_ICP_VOXEL_HB=$hb python scripts/dev/voxel_one.py $n 0.01 2>&1 | grep "points only" | sed "s/^/hb $hb: /"; done; done
