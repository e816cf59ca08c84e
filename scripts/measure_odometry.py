#!/usr/bin/env python3
"""RGB-D odometry (mi_icp_compute_rgbd_odometry) on synthetic frames: ms per call by image size."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import render_rgbd, small_pose
from cupoch_amd.engine import Engine
eng = Engine(0)
pose_b = small_pose(0.02, 0.03)
for (w, h) in ((320, 240), (640, 480), (1280, 960)):
    s = w / 640.0
    K = [525.0 * s, 525.0 * s, (319.5 + 0.5) * s - 0.5, (239.5 + 0.5) * s - 0.5]
    ca, da = render_rgbd(w, h, K, np.eye(4)); cb, db = render_rgbd(w, h, K, pose_b)
    t = [torch.from_numpy(x).cuda() for x in (cb, db, ca, da)]
    for jac in (1, 0, 2):   # 2: ComputeWeightedRGBDOdometry (hybrid term, t-distribution weights)
        f = lambda: eng.compute_rgbd_odometry(t[0], t[1], t[2], t[3], K, None, min(jac, 1), (20, 10, 5), 0.03, 0.0, 6.0,
                                              jac == 2)[:2] + (None,)
        for _ in range(3): f()
        torch.cuda.synchronize(); ts = []
        for _ in range(10):
            t0 = time.perf_counter(); ok, T, info = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ms = float(np.median(ts)) * 1e3
        print(json.dumps({"row": "ComputeRGBDOdometry %dx%d, %s term, iterations (20,10,5)" % (w, h, ("weighted hybrid", "colour", "hybrid")[jac - 2] if jac != 1 else "hybrid"),
                          "ms_per_call": ms, "us_per_iteration": ms * 1e3 / 35, "pose_error": float(np.linalg.norm(T - pose_b)),
                          "motion": float(np.linalg.norm(np.eye(4) - pose_b))}), flush=True)
