#!/bin/bash
# Round 5, run v: config 2 regressed in the late pass (0.023 -> 2.0 ms per iteration): which change?
O=gpurun_out/r05v
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default wantone through; do
  L=$R/cupoch_amd/lib/libmi_icp_$v.so; [ $v = default ] && L=$R/cupoch_amd/lib/libmi_icp.so
  MI_ICP_LIB_PATH=$L timeout 120 python scripts/dev/config2_trace.py 2>&1 | grep -v amdgpu.ids
done | tee $O/config2_trace.txt
