#!/bin/bash
# Round 5, run w: config 2's 60 ms in scripts/measure_configs.py -- with this round's late changes or without (oldish = one-word counter + planes through points)?
O=gpurun_out/r05w
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default oldish default oldish; do
  L=$R/cupoch_amd/lib/libmi_icp_$v.so; [ $v = default ] && L=$R/cupoch_amd/lib/libmi_icp.so
  echo "== $v"
  MI_ICP_LIB_PATH=$L timeout 200 python scripts/measure_configs.py 2>&1 | grep -E "config2|config5|target tree" | cut -c1-200
done | tee $O/configs_ab.txt
