#!/usr/bin/env python3
"""Tracker-side callers of the ICP path: depth frame -> point-cloud pyramid
(PointCloud::CreateFromRGBDImage with normals) and KinfuPipeline::PoseEstimation's
coarse-to-fine RegistrationICP, on synthetic 640x480 frames.  Prints one JSON line per row."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import render_depth, small_pose          # noqa: E402
from cupoch_amd import camera, geometry, kinfu           # noqa: E402

K = [525.0, 525.0, 319.5, 239.5]


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def main():
    eng = geometry.get_engine(0)
    for (w, h) in ((640, 480), (1280, 960), (3840, 2160)):
        s = w / 640.0
        k = [K[0] * s, K[1] * s, (K[2] + 0.5) * s - 0.5, (K[3] + 0.5) * s - 0.5]
        d = torch.from_numpy(render_depth(w, h, k, np.eye(4), holes=0.05)).cuda()
        col = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda")
        ms = timed(lambda: eng.create_from_depth(d, k, None, col, depth_cutoff=6.0, rgbd=True, compute_normals=True))
        ms_d = timed(lambda: eng.create_from_depth(d, k))
        npix = w * h
        # algorithmic bytes: depth 4 + colour 3 read, 36 written per kept pixel (~95 %)
        print(json.dumps({"row": "CreateFromRGBDImage + normals + colours, %dx%d" % (w, h), "ms": ms,
                          "Mpixels_per_s": npix / ms / 1e3, "algorithmic_GBps": npix * (7 + 0.95 * 36) / ms / 1e6,
                          "depth_only_ms": ms_d}))

    levels = 3
    intr = camera.PinholeCameraIntrinsic(640, 480, *K)
    pose_b = small_pose(0.02, 0.03)
    da, db = [], []
    for i in range(levels):
        lv = intr.create_pyramid_level(i)
        da.append(torch.from_numpy(render_depth(lv.width, lv.height, lv.as4(), np.eye(4))).cuda())
        db.append(torch.from_numpy(render_depth(lv.width, lv.height, lv.as4(), pose_b)).cuda())
    for iters in ((10, 10, 10), (20, 20, 20)):
        opt = kinfu.KinfuOption(num_pyramid_levels=levels, depth_cutoff=6.0, distance_threshold=0.03,
                                icp_iterations=iters)
        model = kinfu.point_cloud_pyramid(da, intr, opt)
        out = {}

        def frame_step():
            frame = kinfu.point_cloud_pyramid(db, intr, opt)
            out["T"], _ = kinfu.pose_estimation(opt, np.eye(4, dtype=np.float32), frame, model)
        ms = timed(frame_step, reps=15)
        ms_pc = timed(lambda: kinfu.point_cloud_pyramid(db, intr, opt), reps=15)
        print(json.dumps({"row": "KinFu tracking step: 3-level cloud pyramid + PoseEstimation, 640x480, iterations %s" % (iters,),
                          "ms_per_frame": ms, "frames_per_s": 1e3 / ms, "cloud_pyramid_ms": ms_pc,
                          "points": [len(p.points) for p in model],
                          "pose_error": float(np.linalg.norm(out["T"] - pose_b))}))


if __name__ == "__main__":
    main()
