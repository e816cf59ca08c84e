#!/bin/bash
O=gpurun_out/r04d
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { (cd /tmp && timeout 300 rocprofv3 "$@"); }
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|error" $O/t_gpu.log | tail -3
python scripts/dev/voxel_one.py 2>&1 | grep voxel | tee $O/voxel_new.txt
prof --kernel-trace --stats --output-format csv -d $R/$O/st_voxel_new -o s -- python $R/scripts/dev/voxel_one.py > $O/st_voxel_new.log 2>&1; echo "stats voxel new rc=$?"
python - <<'PY'
import csv,re
for r in csv.DictReader(open("gpurun_out/r04d/st_voxel_new/s_kernel_stats.csv")):
    name=re.sub(r"\(.*","",r["Name"]).replace("void ","").replace("mi::","")
    print("%-40s calls %3s avg %9.1f us  min %8.1f max %8.1f" % (name[:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
timeout 600 python scripts/dev/transient_census.py 2>&1 | grep -E "after|walkers" | tee $O/transient_census.txt
timeout 600 python scripts/measure_latency.py 10000000 2>&1 | grep '^{' | tee $O/latency10m.jsonl | cut -c1-300
