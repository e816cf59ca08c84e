export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_tree_invariants.py tests/test_gpu_seeded.py -x -q 2>&1 | tail -3
cd /tmp
MI_ICP_LINKS_SYNC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sel -o c -- python $R/scripts/measure_latency.py 10000000 > $R/gpurun_out/prof_sel.log 2>&1
cd $R
grep -E "leaf_links" gpurun_out/prof_sel/c_kernel_stats.csv | cut -d, -f1-4 | sed 's/(.*)"//' | cut -c1-100
python scripts/measure_latency.py 10000000 | cut -c60-260
