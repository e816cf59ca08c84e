#!/usr/bin/env python3
"""Reads the PMC passes of scripts/gpu_traffic.sh: calibration factors (reported / known bytes) per
access shape, then the search kernel's FETCH_SIZE / WRITE_SIZE per launch corrected with them.
Writes gpurun_out/nn_traffic.json."""
import collections
import csv
import glob
import json
import re

known = {}
for l in open("gpurun_out/calib_known.txt"):
    m = re.match(r"known_bytes (\w+) (\d+)", l)
    if m:
        known[m.group(1)] = int(m.group(2))


def per_kernel(pattern, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


print("# FETCH_SIZE / WRITE_SIZE calibration (rocprofv3 --pmc, counter unit: KB = 1024 B); 1 GiB buffers")
factor = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = per_kernel("gpurun_out/calib_%s/*counter_collection.csv" % counter, counter)
    for k, v in sorted(agg.items()):
        name = k.split("(")[0]
        if name not in known:
            continue
        rep = sum(v) / len(v) * 1024.0
        if (counter == "WRITE_SIZE") != (name == "calib_write4"):
            continue
        factor[name] = rep / known[name]
        print("%-16s %-10s reported %13.0f B  known %13d B  reported/known = %.4f  (%d launches)"
              % (name, counter, rep, known[name], rep / known[name], len(v)))

fetch = per_kernel("gpurun_out/nnpmc_FETCH_SIZE/*counter_collection.csv", "FETCH_SIZE")
write = per_kernel("gpurun_out/nnpmc_WRITE_SIZE/*counter_collection.csv", "WRITE_SIZE")
key = [k for k in fetch if "nn_packet_kernel<true, false, false>" in k or "nn_packet_kernel<true, false>" in k or "nn_packet_kernel<true,false>" in k]
out = {}
if key:
    k = key[0]
    f_kb = sum(fetch[k]) / len(fetch[k])
    w_kb = sum(write[k]) / len(write[k]) if k in write else float("nan")
    # the seeded search kernel's reads are: 4 B/lane SoA (source x, y, z, previous match: 16 B/point),
    # 16 B/lane leaf lines shared by ~8 lanes, 2 x 16 B/lane region records; on this workload no tree
    # records and no list chunks.  If the calibrated shapes agree on one factor it is applied as is;
    # otherwise the factors are weighted by the shapes' algorithmic byte shares.
    shares = {"calib_soa4": 16.0, "calib_leaf16": 16.0 / 0.6, "calib_pair32": 4.0 / 0.6}   # bytes per source point
    tot = sum(shares.values())
    fsum = sum(shares[s] / factor.get(s, 1.0) for s in shares) / tot        # mean of 1/factor, byte-weighted
    fetch_bytes = f_kb * 1024.0 * fsum
    write_bytes = w_kb * 1024.0 / factor.get("calib_write4", 1.0)
    print("\n# search kernel (seeded), 10M-vs-10M bench, per launch (average of %d launches)" % len(fetch[k]))
    print("FETCH_SIZE reported %.0f B, corrected x%.4f -> %.0f B" % (f_kb * 1024, fsum, fetch_bytes))
    print("WRITE_SIZE reported %.0f B, corrected /%.4f -> %.0f B" % (w_kb * 1024, factor.get("calib_write4", 1.0), write_bytes))
    print("HBM bytes per launch = %.0f (algorithmic 20 N_s + 20 N_t = 400000000)" % (fetch_bytes + write_bytes))
    out = {"kernel": "mi::nn_packet_kernel<true,false>", "points": 10000000, "n_gpus": 1,
           "source": "scripts/gpu_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate "
                     "passes) of python bench.py --steps 6 --warmup 2; calibration scripts/ubench/fetch_calib.hip",
           "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb, "calibration_reported_over_known": factor,
           "fetch_correction": fsum, "hbm_bytes_per_launch": int(fetch_bytes + write_bytes),
           "note": "FETCH_SIZE x %.3f (byte-weighted over the kernel's access shapes, calibrated on 1 GiB of known "
                   "bytes each: profiles/r06_fetch_calibration.txt) + WRITE_SIZE / %.3f" % (fsum, factor.get("calib_write4", 1.0))}
    json.dump(out, open("gpurun_out/nn_traffic.json", "w"), indent=1)
else:
    print("no nn_packet_kernel<true,false> rows found")
