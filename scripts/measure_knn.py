#!/usr/bin/env python3
"""KDTreeFlann-style k-NN search throughput (mi_icp_search_knn): 2M queries against 10M points."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd.engine import Engine
eng = Engine(0)
src, tgt, nrm, T, md = synth(10_000_000)
d_t, d_q = torch.from_numpy(tgt).cuda(), torch.from_numpy(src[:2_000_000]).cuda()
eng.set_target(d_t)
CASES = ((1, 0.0), (8, 0.0), (30, 0.0), (30, 0.01))
if len(sys.argv) > 1:
    CASES = [tuple(float(x) if "." in x else int(x) for x in a.split(",")) for a in sys.argv[1:]]
for k, r in CASES:
    eng.search_knn(d_q, k, r); torch.cuda.synchronize()
    t0 = time.perf_counter(); found, idx, d2 = eng.search_knn(d_q, k, r); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"row": "KDTreeFlann search, 2M queries vs 10M points", "knn": k, "radius": r, "ms": dt * 1e3,
                      "Mqueries_per_s": 2.0 / dt, "found": found}), flush=True)
