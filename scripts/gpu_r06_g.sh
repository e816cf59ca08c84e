#!/bin/bash
# Round 6: where a small cloud's set_target / set_source go (kernel trace, 20k points)
O=gpurun_out/r06g
mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
timeout 120 python scripts/dev/small_build_trace.py 20000 2>&1 | grep -v amdgpu.ids
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o sb -- python $R/scripts/dev/small_build_trace.py 20000 > $R/$O/prof.log 2>&1; echo "prof rc=$?"
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r06g/prof/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("kernels per (set_target + set_source): %.1f, kernel time per pair %.1f us" % (calls / 55.0, tot / 55.0 / 1e3))
for r in rows[:25]:
    print("%-58s calls/pair %5.1f avg %7.1f us  per pair %6.1f us" % (r["Name"][:58], int(r["Calls"]) / 55.0, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 55.0 / 1e3))
PY
