#!/bin/bash
# Round 5, run l: where kd_build_groups' 0.87 ms go -- variant libraries without the in-group sort, without the attribute
# stores, with coherent instead of gathered loads (timing only: their trees are wrong).
O=gpurun_out/r05l
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default nosort nowrite coh nosortnowrite default; do
  L=$R/cupoch_amd/lib/libmi_icp_$v.so; [ $v = default ] && L=$R/cupoch_amd/lib/libmi_icp.so
  (cd /tmp && MI_ICP_LIB_PATH=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/st_$v -o s -- python $R/scripts/dev/build_dissect.py) 2>&1 | grep "set_target ms"
  echo "  $v: kd_build_groups avg us: $(grep kd_build_groups $O/st_$v/*kernel_stats.csv | awk -F, '{print $(NF-4)/1000}')"
done 2>&1 | tee $O/dissect.txt
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete
