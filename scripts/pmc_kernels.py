#!/usr/bin/env python3
"""Per-kernel digest of rocprofv3 --pmc passes: python scripts/pmc_kernels.py <glob of *_counter_collection.csv> [name-substring ...]
Averages every counter per launch for the kernels whose name contains one of the substrings (all if none)."""
import collections, csv, glob, sys
files = sorted(glob.glob(sys.argv[1]))
pats = sys.argv[2:]
for f in files:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mi::", "")
        if pats and not any(p in name for p in pats):
            continue
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        v = agg[k]
        print("%-44s launches %4d  " % (k[:44], max(len(x) for x in v.values())) + "  ".join("%s %.4g" % (c, sum(x) / len(x)) for c, x in sorted(v.items())))
