#!/bin/bash
# Round 5, run p: is the binary descent accurate?  (test + census of the cold call's second search from either kind of seed)
O=gpurun_out/r05p
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_tree_invariants.py -x -q -k locate 2>&1 | tail -15 | tee $O/test_locate.txt
MI_ICP_NO_LINKS=1 timeout 300 python scripts/dev/cold_census.py 10000000 2>&1 | grep -v amdgpu.ids | tee $O/cold_census.txt
