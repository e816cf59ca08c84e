#!/bin/bash
# Round 6: the KinFu tracking step -- GPU time against wall time (kernel trace + memory copies)
O=gpurun_out/r06i
mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/prof -o kf -- python $R/scripts/measure_kinfu.py > $R/$O/prof.log 2>&1; echo "prof rc=$?"
cd $R
grep KinFu $O/prof.log | cut -c1-200
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r06i/prof/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
print("total kernel ms", sum(float(r["TotalDurationNs"]) for r in rows) / 1e6, "calls", sum(int(r["Calls"]) for r in rows))
for r in rows[:16]:
    print("%-60s calls %6s avg %8.1f us total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
for g in glob.glob("gpurun_out/r06i/prof/*memory_copy_stats.csv"):
    for r in csv.DictReader(open(g)):
        print("copy", r)
PY
