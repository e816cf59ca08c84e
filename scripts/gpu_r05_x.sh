#!/bin/bash
# Round 5, run x: the k-NN walk in two rounds (cubes cut to the packet's mean bound first) against one round
# (-DMI_AB_KNN_ONE_ROUND), same box; the k-NN / normals / colour tests first.
O=gpurun_out/r05x
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kdtree.py tests/test_gpu_colored.py tests/test_gpu_baseline_configs.py tests/test_gpu_cpp.py tests/test_gpu_python_api.py tests/test_gpu_robustness.py tests/test_io_and_real_data.py tests/test_pybind_module.py tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_depth.py -m gpu -q -x --timeout=600 > $O/tests_full.log 2>&1; grep -E "passed|failed|error" $O/tests_full.log | tail -3 | tee $O/tests.txt
{
for v in oneround default oneround default; do
  L=$R/cupoch_amd/lib/libmi_icp_$v.so; [ $v = default ] && L=$R/cupoch_amd/lib/libmi_icp.so
  echo "== $v"
  MI_ICP_LIB_PATH=$L timeout 200 python scripts/measure_normals_10m.py 2>&1 | grep normals
  MI_ICP_LIB_PATH=$L timeout 200 python scripts/measure_knn.py 1,0.0 8,0.0 30,0.0 30,0.01 64,0.0 100,0.0 2>&1 | grep '^{' | cut -c50-140
done
} 2>&1 | tee $O/knn_ab.txt
