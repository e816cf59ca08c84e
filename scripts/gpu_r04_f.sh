#!/bin/bash
O=gpurun_out/r04f
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { (cd /tmp && timeout 300 rocprofv3 "$@"); }
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|error" $O/t_gpu.log | tail -3
timeout 600 python scripts/measure_latency.py 10000000 2>&1 | grep '^{' | tee $O/latency10m.jsonl | cut -c1-300
prof --kernel-trace --stats --output-format csv -d $R/$O/st_cold -o s -- python $R/scripts/measure_latency.py 10000000 > $O/st_cold.log 2>&1; echo "stats cold rc=$?"
python - <<'PY'
import csv,re
for r in csv.DictReader(open("gpurun_out/r04f/st_cold/s_kernel_stats.csv")):
    name=re.sub(r"\(.*","",r["Name"]).replace("void ","").replace("mi::","")
    print("%-40s calls %3s avg %9.1f us  min %8.1f max %8.1f tot %9.1f" % (name[:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
