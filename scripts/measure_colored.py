#!/usr/bin/env python3
"""Colored ICP row: colour-gradient initialisation and
iterations/s on a 2M-point textured surface.  One JSON object per line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cupoch_amd import _lib                  # noqa: E402
from cupoch_amd.engine import Engine         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
scale = 0.7 * np.sqrt(n)                      # ~1 unit between neighbouring points
rng = np.random.default_rng(0)
xy = rng.random((n, 2))
z = 0.1 * np.sin(4 * xy[:, 0]) * np.cos(3 * xy[:, 1])
tgt = (np.stack([xy[:, 0], xy[:, 1], z], 1) * scale).astype(np.float32)
inten = 0.5 + 0.4 * np.sin(40 * xy[:, 0]) * np.cos(30 * xy[:, 1])
col = np.stack([inten, 0.9 * inten, 0.8 * inten], 1).astype(np.float32)
a = 2e-4
T = np.eye(4)
T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
T[:3, 3] = [0.3, -0.2, 0.05]
Ti = np.linalg.inv(T)
src = (tgt.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
perm = rng.permutation(n)
src, scol = np.ascontiguousarray(src[perm]), np.ascontiguousarray(col[perm])

eng = Engine(0)
d_tgt, d_src = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
d_col, d_scol = torch.from_numpy(col).cuda(), torch.from_numpy(scol).cuda()
t0 = time.perf_counter()
nrm = eng.estimate_normals_knn(d_tgt, 20)
torch.cuda.synchronize()
t_nrm = time.perf_counter() - t0
nrm = torch.where(nrm[:, 2:3] < 0, -nrm, nrm).contiguous()
eng.set_target(d_tgt, nrm)
eng.set_source(d_src)
eng.set_target_colors(d_col)
eng.set_source_colors(d_scol)
max_dist = 2.0
ts = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.compute_color_gradients(2 * max_dist, 30, want_output=False)
    eng.synchronize()
    ts.append(time.perf_counter() - t0)
t_grad = float(np.median(ts))
print(json.dumps({"row": "InitializePointCloudForColoredICP (radius 2*max_dist, max_nn 30)", "n": n,
                  "ms": t_grad * 1e3, "Mpts_per_s": n / t_grad / 1e6,
                  "estimate_normals_knn20_ms": t_nrm * 1e3}), flush=True)
eng.set_profiling(True)
eng.icp_begin(_lib.EST_COLORED, max_dist, None, -1.0)
eng.icp_iterate(3)
p0 = eng.get_profile()
torch.cuda.synchronize()
t0 = time.perf_counter()
res = eng.icp_iterate(30)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
p1 = eng.get_profile()
Tg = np.array(res.transformation, np.float32).reshape(4, 4).T
print(json.dumps({"row": "colored ICP %d-vs-%d (lambda 0.968)" % (n, n), "it_per_s": 30 / dt,
                  "ms_per_iter": dt / 30 * 1e3, "nn_ms": (p1["nn_ms"] - p0["nn_ms"]) / 30,
                  "reduce_ms": (p1["reduce_ms"] - p0["reduce_ms"]) / 30,
                  # per correspondence: 12 B source + 4 B intensity + 4 B index, 16 B target line share,
                  # 16 B normal+intensity, 16 B gradient
                  "reduce_algorithmic_GBps": 68.0 * n / ((p1["reduce_ms"] - p0["reduce_ms"]) / 30 * 1e-3) / 1e9,
                  "fitness": res.fitness, "T_err_vs_gt": float(np.linalg.norm(Tg - T)),
                  "motion": float(np.linalg.norm(T - np.eye(4)))}), flush=True)
eng.close()
