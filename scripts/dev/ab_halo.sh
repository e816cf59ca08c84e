#!/bin/bash
# same-box A/B of the halo build kernels' durations (rocprofv3 kernel stats of scripts/dev/halo_build_time.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for which in ${AB_LIBS:-prev new prev new}; do
  echo "== $which"
  (cd /tmp && MI_ICP_LIB_PATH=$R/cupoch_amd/lib/ab_$which.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abh_$which -o s -- python $R/scripts/dev/halo_build_time.py > $R/gpurun_out/abh_$which.log 2>&1)
  tail -2 gpurun_out/abh_$which.log
  python - <<PY
import csv,glob
for f in glob.glob("gpurun_out/abh_$which/**/s_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "leaf_halo" in r["Name"]: print("  %-20s calls %s avg %.1f us" % (r["Name"].split("(")[0][-20:], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
