#!/usr/bin/env python3
"""Measures every row of SURVEY.md section 8 that bench.py's single JSON line does
not cover (BASELINE.json configs 2 and 5, VoxelDownSample, LBVH build,
EstimateNormals, Transform).  One JSON object per line; kept under profiles/."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth                      # noqa: E402
from cupoch_amd import _lib                  # noqa: E402
from cupoch_amd.engine import Engine         # noqa: E402

eng = Engine(0)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def emit(**kw):
    print(json.dumps(kw), flush=True)


def gpu(a):
    return torch.from_numpy(a).cuda()


# ---- config 2: 1M-vs-1M -> VoxelDownSample(0.02) both -> point-to-plane, r = 0.04
src, tgt, nrm, T_gt, _ = synth(1_000_000)
d_src, d_tgt, d_nrm = gpu(src), gpu(tgt), gpu(nrm)
src_nrm = (nrm @ np.linalg.inv(T_gt)[:3, :3].T.astype(np.float32))
t_vox = timed(lambda: eng.voxel_downsample(d_tgt, 0.02, d_nrm))
vt, vn, _ = eng.voxel_downsample(d_tgt, 0.02, d_nrm)
vs, _, _ = eng.voxel_downsample(d_src, 0.02)
m = len(vt)
emit(row="VoxelDownSample", n=1_000_000, voxel=0.02, voxels=m, ms=t_vox * 1e3,
     algorithmic_GBps=12 * 2 * (1_000_000 + m) / t_vox / 1e9)
eng.set_target(vt, vn)
eng.set_source(vs)
eng.icp_begin(_lib.EST_POINT_TO_PLANE, 0.04, None, -1.0)
eng.icp_iterate(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = eng.icp_iterate(30)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
T = np.array(res.transformation, np.float32).reshape(4, 4).T
emit(row="config2: 1M->voxel(0.02)->pt2pl ICP r=0.04", n_source=len(vs), n_target=m,
     it_per_s=30 / dt, ms_per_iter=dt / 30 * 1e3, fitness=res.fitness, rmse=res.inlier_rmse,
     T_err_vs_gt=float(np.linalg.norm(T - T_gt)))

# ---- config 5: GICP 5M-vs-5M, covariances R diag(1e-3,1,1) R^T from the normals
n5 = 5_000_000
src, tgt, nrm, T_gt, max_dist = synth(n5)
d_src, d_tgt, d_nrm = gpu(src), gpu(tgt), gpu(nrm)
perm = np.random.Generator(np.random.PCG64(44)).permutation(n5)
src_nrm = np.ascontiguousarray((nrm @ np.linalg.inv(T_gt)[:3, :3].T.astype(np.float32))[perm])
t_cov = timed(lambda: eng.covariances_from_normals(d_nrm, 1e-3), 3)
tcov = eng.covariances_from_normals(d_nrm, 1e-3)
scov = eng.covariances_from_normals(gpu(src_nrm), 1e-3)
emit(row="covariances_from_normals", n=n5, ms=t_cov * 1e3, GBps=(12 + 36) * n5 / t_cov / 1e9)
eng.set_target(d_tgt, d_nrm, tcov)      # first call at this size: buffer growth (hipMalloc), not timed
eng.set_source(d_src, None, scov)
eng.synchronize()
t0 = time.perf_counter()
eng.set_target(d_tgt, d_nrm, tcov)
eng.set_source(d_src, None, scov)
eng.synchronize()
emit(row="tree build + source staging (GICP, with covariances)", n=n5, ms=(time.perf_counter() - t0) * 1e3)
eng.set_profiling(True)
eng.icp_begin(_lib.EST_GENERALIZED, max_dist, None, -1.0)
eng.icp_iterate(3)
p0 = eng.get_profile()
torch.cuda.synchronize()
t0 = time.perf_counter()
res = eng.icp_iterate(30)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
p1 = eng.get_profile()
T = np.array(res.transformation, np.float32).reshape(4, 4).T
emit(row="config5: GICP 5M-vs-5M", it_per_s=30 / dt, ms_per_iter=dt / 30 * 1e3,
     nn_ms=(p1["nn_ms"] - p0["nn_ms"]) / 30, reduce_ms=(p1["reduce_ms"] - p0["reduce_ms"]) / 30,
     algorithmic_GBps=(132 * n5 + 20 * n5) * 30 / dt / 1e9, fitness=res.fitness,
     T_err_vs_gt=float(np.linalg.norm(T - T_gt)))
eng.set_profiling(False)

# ---- one-time costs at 10M: LBVH build, Transform, EstimateNormals
n = 10_000_000
src, tgt, nrm, T_gt, max_dist = synth(n)
d_src, d_tgt, d_nrm = gpu(src), gpu(tgt), gpu(nrm)
t = timed(lambda: (eng.set_target(d_tgt, d_nrm)), 3)
emit(row="target tree build (set_target, with normals; kd cells + 8-ary records)", n=n, ms=t * 1e3, algorithmic_GBps=36 * n / t / 1e9)
t = timed(lambda: (eng.set_source(d_src)), 3)
emit(row="source Morton staging (set_source)", n=n, ms=t * 1e3)
pts = d_src.clone()
t = timed(lambda: eng.transform(T_gt, pts), 5)
emit(row="PointCloud::Transform (points)", n=n, ms=t * 1e3, GBps=24 * n / t / 1e9)
t = timed(lambda: eng.voxel_downsample(d_tgt, 0.01, d_nrm), 3)
vt, _, _ = eng.voxel_downsample(d_tgt, 0.01, d_nrm)
emit(row="VoxelDownSample", n=n, voxel=0.01, voxels=len(vt), ms=t * 1e3,
     algorithmic_GBps=24 * (n + len(vt)) / t / 1e9)
n2 = 2_000_000
t = timed(lambda: eng.estimate_normals_knn(d_tgt[:n2], 30), 2)
emit(row="EstimateNormals(KNN 30)", n=n2, ms=t * 1e3, Mpts_per_s=n2 / t / 1e6)
t = timed(lambda: eng.estimate_normals_knn(d_tgt[:n2], 20), 2)
emit(row="EstimateNormals(KNN 20)", n=n2, ms=t * 1e3, Mpts_per_s=n2 / t / 1e6)
