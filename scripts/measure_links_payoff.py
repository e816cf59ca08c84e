#!/usr/bin/env python3
"""Do the leaf neighbour lists pay for themselves?  A whole RegistrationICP call (build + staging + loop) on
noisy, partially overlapping clouds (sigma = 0.15 spacings, 60 % overlap) at several sizes and iteration
budgets; run once as is and once with MI_ICP_NO_LINKS=1."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd import _lib
from cupoch_amd.engine import Engine
eng = Engine(0)
for n in [int(a) for a in sys.argv[1:]] or [100_000, 307_200, 1_000_000, 3_000_000]:
    src, tgt, nrm, T_gt, max_dist = synth(n)
    rng = np.random.default_rng(5)
    keep = rng.random(n) < 0.6
    s = float(n) ** (-1.0 / 3.0)
    noisy = (src[keep] + rng.normal(0.0, 0.15 * s, (int(keep.sum()), 3))).astype(np.float32)
    d_src, d_tgt, d_nrm = torch.from_numpy(noisy).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda()
    for iters in (10, 30):
        def call():
            eng.set_target(d_tgt, d_nrm)
            eng.set_source(d_src)
            return eng.registration_icp(_lib.EST_POINT_TO_PLANE, max_dist, None, 0.0, 0.0, iters, -1.0)
        call(); call()
        ts = []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter(); res = call(); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(json.dumps({"n": n, "iterations": iters, "links": "off" if os.environ.get("MI_ICP_NO_LINKS") else "on",
                          "call_ms": round(float(np.median(ts)) * 1e3, 3), "fitness": round(res.fitness, 4)}), flush=True)
eng.close()
