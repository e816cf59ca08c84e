#!/bin/bash
# Round 5, run n: what the finishing block's "solve" span (4.2 us) is made of -- variants without the solve, without the re-location sizing.
O=gpurun_out/r05n
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default nosolve nosizing default; do
  L=$R/cupoch_amd/lib/libmi_icp_$v.so; [ $v = default ] && L=$R/cupoch_amd/lib/libmi_icp.so
  echo "== $v"
  MI_ICP_LIB_PATH=$L timeout 300 python scripts/measure_step_breakdown.py 2>&1 | grep -E "^N = |solve|rows totalled|state written|step end"
done 2>&1 | tee $O/solve_span.txt
