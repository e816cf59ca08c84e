#!/bin/bash
# Round 4: voxel path iteration (full gpu suite + timing + kernel stats)
O=gpurun_out/r04c
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { (cd /tmp && timeout 300 rocprofv3 "$@"); }
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -5
python scripts/dev/voxel_one.py 2>&1 | grep voxel | tee $O/voxel_new.txt
python scripts/dev/voxel_one.py 1000000 0.02 2>&1 | grep voxel | tee -a $O/voxel_new.txt
prof --kernel-trace --stats --output-format csv -d $R/$O/st_voxel_new -o s -- python $R/scripts/dev/voxel_one.py > $O/st_voxel_new.log 2>&1; echo "stats voxel new rc=$?"
python - <<'PY'
import csv,re
for r in csv.DictReader(open("gpurun_out/r04c/st_voxel_new/s_kernel_stats.csv")):
    name=re.sub(r"\(.*","",r["Name"]).replace("void ","").replace("mi::","")
    print("%-40s calls %3s avg %9.1f us  min %8.1f max %8.1f" % (name[:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
