#!/bin/bash
# Round 6: where the group search's first pass spent its time (kernel trace of scripts/dev/gs_first_pass.py, both removed with
# the experiment: git show 5f03866), and which inputs abort the search
O=gpurun_out/r06d
mkdir -p $O
export TMPDIR=/tmp
for k in far; do
  timeout 120 python scripts/dev/nan_queries.py $k 2>&1 | grep -v amdgpu.ids | tail -1
  MI_ICP_NO_GROUP_SEARCH=1 timeout 120 python scripts/dev/nan_queries.py $k 2>&1 | grep -v amdgpu.ids | tail -1
done
timeout 300 python scripts/dev/gs_first_pass.py 2>&1 | grep -v amdgpu.ids
R=$(pwd)
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o gs -- python $R/scripts/dev/gs_first_pass.py > $R/$O/prof.log 2>&1; echo "prof rc=$?"
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo $f
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %5s avg %10.1f us total %8.2f ms %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
