#!/usr/bin/env python3
"""Experiment: how much of the traversal is upper-level Morton misalignment?
Target A: uniform random (the bench workload).  Target B: the same number of points
stratified so that every 4096-point Morton run IS an octree cell (4096 cells x 4096
points, 4 % gaps between cells so that the quantisation cannot leak points across).
Same source construction for both.  Prints nodes/leaves per packet and ms/iteration."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cupoch_amd import _lib                  # noqa: E402
from cupoch_amd.engine import Engine         # noqa: E402

G = 16                                        # cells per axis
n = G ** 3 * 4096
rng = np.random.Generator(np.random.PCG64(42))


def make(stratified):
    if stratified:
        cell = np.stack(np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij"), -1).reshape(-1, 3)
        u = rng.random((G ** 3, 4096, 3), dtype=np.float32)
        tgt = ((cell[:, None, :] + 0.02 + 0.96 * u) / G).reshape(-1, 3).astype(np.float32)
        tgt = tgt[rng.permutation(n)]
    else:
        tgt = rng.random((n, 3), dtype=np.float32)
    nrm = rng.standard_normal((n, 3), dtype=np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    s = float(n) ** (-1.0 / 3.0)
    ang = 0.2 * s
    ax = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    t = 0.2 * s * np.array([1.0, -1.0, 1.0]) / np.sqrt(3.0)
    src = (tgt.astype(np.float64) @ R + (-R.T @ t)).astype(np.float32)   # R^T applied: p R = R^T p
    src = np.ascontiguousarray(src[rng.permutation(n)])
    T = np.eye(4, dtype=np.float32)
    T[:3, :3], T[:3, 3] = R, t
    return src, tgt, nrm, T, 2.0 * s


eng = Engine(0)
for strat in (False, True):
    src, tgt, nrm, T_gt, max_dist = make(strat)
    eng.set_target(torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda())
    eng.set_source(torch.from_numpy(src).cuda())
    eng.set_profiling(True)
    eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, None, -1.0)
    eng.icp_iterate(5)
    p0 = eng.get_profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = eng.icp_iterate(30)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p1 = eng.get_profile()
    T = np.array(res.transformation, np.float32).reshape(4, 4).T
    out = (C.c_uint64 * 4)()
    a = np.ascontiguousarray(T.T)
    eng._chk(eng._L.mi_icp_debug_nn_stats(eng._ctx, a.ctypes.data_as(C.c_void_p), float(max_dist), 1, out))
    print("%s n=%d: nodes/packet %.1f leaves/packet %.1f | %.3f ms/iter (nn %.3f) | T err %.2g fitness %.4f"
          % ("stratified (aligned runs)" if strat else "uniform random          ", n, out[0] / out[2],
             out[1] / out[2], dt / 30 * 1e3, (p1["nn_ms"] - p0["nn_ms"]) / 30, np.linalg.norm(T - T_gt),
             res.fitness), flush=True)
eng.close()
