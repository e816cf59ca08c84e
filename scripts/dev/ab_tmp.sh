cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_voxel_dense.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_scale.py -x -q -m gpu -k "voxel" 2>&1 | tail -2
for v in 0.001 0.003 0.005; do python scripts/dev/voxel_one.py 10000000 $v 2>&1 | grep voxel; done
