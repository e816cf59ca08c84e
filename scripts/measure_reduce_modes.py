#!/usr/bin/env python3
"""reduce_kernel at 10M: full system (MODE 0, ~100 VGPRs, 4 waves/SIMD) vs RMSE only (MODE 1, few VGPRs)."""
import sys, os, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd import _lib
from cupoch_amd.engine import Engine
n = 10_000_000
src, tgt, nrm, T_gt, max_dist = synth(n)
eng = Engine(0)
eng.set_target(torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda())
eng.set_source(torch.from_numpy(src).cuda())
eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, T_gt, -1.0)   # match-ordered source, correspondences under T_gt
eng.set_profiling(True)
for name, fn in (("system (MODE 0)", lambda: eng.compute_system(_lib.EST_POINT_TO_PLANE, T_gt)),
                 ("rmse only (MODE 1)", lambda: eng.compute_rmse(_lib.EST_POINT_TO_PLANE, T_gt))):
    fn()
    p0 = eng.get_profile()
    for _ in range(10):
        fn()
    p1 = eng.get_profile()
    print(json.dumps({"reduce": name, "ms": (p1["reduce_ms"] - p0["reduce_ms"]) / max(p1["reduce_launches"] - p0["reduce_launches"], 1)}))
