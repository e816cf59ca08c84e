#!/usr/bin/env python3
"""BASELINE config 1: CPU point-to-point ICP, 100k-vs-100k synthetic cloud, 30 iterations (plumbing; no GPU).
The reference's methodology (examples/python/basic/benchmarks.py:50-83: wall clock around registration_icp)
on the oracle -- the CPU restatement of the reference's loop (Open3D is not installable here); all host threads
and one thread (README.md:124 quotes the comparison single-threaded)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
src, tgt, nrm, T_gt, max_dist = synth(n)
rows = {}
import subprocess
for threads in ("1", "8", "32", "128"):
    os.environ["OMP_NUM_THREADS"] = threads
    code = ("import sys,time,json,numpy as np; sys.path.insert(0,%r); from bench import synth; from oracle import oracle as orc;"
            "src,tgt,nrm,T,md=synth(%d); orc.registration_icp(src[:1000],tgt[:1000],md);ts=[]\n"
            "for _ in range(5):\n t0=time.perf_counter(); r=orc.registration_icp(src,tgt,md,relative_fitness=0.0,relative_rmse=0.0,max_iteration=30); ts.append(time.perf_counter()-t0)\n"
            "print(json.dumps({'s':float(np.median(ts)),'iters':int(r.iterations),'fitness':float(r.fitness),'rmse':float(r.inlier_rmse),"
            "'T_err':float(np.linalg.norm(r.transformation-T)),'threads':orc.num_threads()}))" % (ROOT, n))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ), timeout=1800)
    rows[threads] = json.loads(out.stdout.strip().splitlines()[-1])
best = min(rows, key=lambda k: rows[k]["s"])
a, o = rows[best], rows["1"]
print(json.dumps({"row": "BASELINE config 1: CPU point-to-point ICP (oracle port), %d-vs-%d, 30 iterations, median of 5 whole calls "
                         "(kd-tree build + 31 searches + 30 Kabsch steps)" % (n, n),
                  "seconds_per_call": round(a["s"], 4), "iterations_per_s": round(30.0 / a["s"], 2), "threads": int(best),
                  "seconds_per_call_by_threads": {k: round(v["s"], 4) for k, v in rows.items()},
                  "one_thread_seconds_per_call": round(o["s"], 4), "one_thread_iterations_per_s": round(30.0 / o["s"], 3),
                  "iterations": a["iters"], "fitness": a["fitness"], "inlier_rmse": a["rmse"], "T_err_fro_vs_ground_truth": a["T_err"]}))
