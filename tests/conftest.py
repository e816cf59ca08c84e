import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_tests.json")) as f:
        return json.load(f)


def make_pair(n, seed=42, rot_scale=0.2, noise=0.0, permute=True):
    """BASELINE.md section 3 synthetic pair: target U[0,1)^3, unit normals,
    source = T_gt^-1 * target (permuted).  Returns dict of float32 arrays."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tgt = rng.random((n, 3), dtype=np.float32)
    nrm = np.random.Generator(np.random.PCG64(seed + 1)).standard_normal((n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    s = float(n) ** (-1.0 / 3.0)
    ang = rot_scale * s
    ax = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    t = rot_scale * s * np.array([1.0, -1.0, 1.0]) / np.sqrt(3.0)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    Tinv = np.linalg.inv(T)
    src = (tgt.astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3]).astype(np.float32)
    src_nrm = (nrm.astype(np.float64) @ Tinv[:3, :3].T).astype(np.float32)
    if noise > 0:
        src += (np.random.Generator(np.random.PCG64(seed + 3)).standard_normal((n, 3)) *
                noise * s).astype(np.float32)
    if permute:
        perm = np.random.Generator(np.random.PCG64(seed + 2)).permutation(n)
        src, src_nrm = src[perm], src_nrm[perm]
    return dict(src=np.ascontiguousarray(src), tgt=tgt, tgt_nrm=nrm,
                src_nrm=np.ascontiguousarray(src_nrm), T_gt=T.astype(np.float32),
                spacing=s, max_dist=2.0 * s)


def make_colored(n=20000, seed=0, scale=100.0, planar=False):
    """Textured surface for colored-ICP cases: z = f(x, y) over a scale x scale patch
    (or the plane z = 0), smooth intensity texture, and a small ground-truth motion.
    `scale` ~ 100 keeps the reference's fp32 AtA.inverse() of the colour-gradient fit
    well conditioned ((nn-1)^2 nt nt^T vs sum v v^T)."""
    rng = np.random.default_rng(seed)
    xy = rng.random((n, 2))
    z = np.zeros(n) if planar else 0.1 * np.sin(4 * xy[:, 0]) * np.cos(3 * xy[:, 1])
    tgt = (np.stack([xy[:, 0], xy[:, 1], z], 1) * scale).astype(np.float32)
    q = tgt / scale
    i = 0.5 + 0.4 * np.sin(9 * q[:, 0]) * np.cos(7 * q[:, 1])
    col = np.stack([i, i * 0.9, np.minimum(i * 1.1, 1.0)], 1).astype(np.float32)
    a = 0.01
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    T[:3, 3] = np.array([0.004, -0.003, 0.0 if planar else 0.001]) * scale
    return tgt, col, T


def colored_fragment_pair():
    """Two interleaved samplings of the reference's coloured RGB-D fragment
    (tests/golden/frag115_every2nd.npz): target = even vertices, source = odd vertices
    moved by the inverse of a known T."""
    d = np.load(os.path.join(ROOT, "tests", "golden", "frag115_every2nd.npz"))
    pts = np.ascontiguousarray(d["points"])
    col = (d["colors"].astype(np.float32) / np.float32(255.0)).astype(np.float32)
    ang = 0.03
    ax = np.array([0.2, 1.0, 0.3])
    ax /= np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    T[:3, 3] = [0.02, -0.015, 0.01]
    Ti = np.linalg.inv(T)
    src = (pts[1::2].astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    return (np.ascontiguousarray(src), np.ascontiguousarray(col[1::2]), np.ascontiguousarray(pts[0::2]),
            np.ascontiguousarray(col[0::2]), T.astype(np.float32))


def render_depth(width, height, K4, cam_pose, holes=0.0, seed=0, scale=1.0):
    """Analytic depth frame (z in the camera frame, float32) of a box room
    [-2,2] x [-1.5,1.5] x [-0.5,4] with two spheres inside, seen from cam_pose
    (camera -> world, 4x4).  holes: fraction of pixels zeroed at random; scale: the
    scene's unit (100 = centimetres)."""
    fx, fy, cx, cy = [float(v) for v in K4]
    P = np.asarray(cam_pose, np.float64).reshape(4, 4)
    v, u = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    dirs_c = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)        # z = 1 along the axis
    dirs = dirs_c @ P[:3, :3].T
    o = P[:3, 3]
    t_best = np.full((height, width), np.inf)
    lo, hi = np.array([-2.0, -1.5, -0.5]) * scale, np.array([2.0, 1.5, 4.0]) * scale
    for ax in range(3):
        for bound in (lo[ax], hi[ax]):
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (bound - o[ax]) / dirs[..., ax]
            hit = o + t[..., None] * dirs
            ok = (t > 1e-6 * scale)
            for b in range(3):
                if b != ax:
                    ok &= (hit[..., b] >= lo[b] - 1e-9 * scale) & (hit[..., b] <= hi[b] + 1e-9 * scale)
            t_best = np.where(ok & (t < t_best), t, t_best)
    for c, r in ((np.array([0.6, 0.4, 2.2]) * scale, 0.55 * scale), (np.array([-0.9, -0.3, 2.8]) * scale, 0.7 * scale)):
        oc = o - c
        a = (dirs * dirs).sum(-1)
        b = 2.0 * (dirs * oc).sum(-1)
        cc = (oc * oc).sum() - r * r
        disc = b * b - 4 * a * cc
        with np.errstate(invalid="ignore"):
            t = (-b - np.sqrt(disc)) / (2 * a)
        ok = (disc > 0) & (t > 1e-6 * scale)
        t_best = np.where(ok & (t < t_best), t, t_best)
    depth = np.where(np.isfinite(t_best), t_best, 0.0).astype(np.float32)         # t scales z = 1
    if holes > 0:
        rng = np.random.default_rng(seed)
        depth[rng.random(depth.shape) < holes] = 0.0
    return depth


def small_pose(angle=0.02, shift=0.03):
    """camera -> world pose: a small rotation about (1,2,3) and a translation"""
    ax = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(angle) * Kx + (1 - np.cos(angle)) * (Kx @ Kx)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = shift * np.array([1.0, -1.0, 0.5]) / 1.5
    return T.astype(np.float32)


def render_rgbd(width, height, K4, cam_pose, holes=0.0, seed=0):
    """(intensity, depth) of the render_depth scene: the intensity is a smooth world-space
    texture, so that two frames of the same surface point agree."""
    depth = render_depth(width, height, K4, cam_pose, holes=holes, seed=seed)
    fx, fy, cx, cy = [float(v) for v in K4]
    v, u = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    z = depth.astype(np.float64)
    pc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], -1)
    P = np.asarray(cam_pose, np.float64).reshape(4, 4)
    pw = pc @ P[:3, :3].T + P[:3, 3]
    tex = 0.5 + 0.2 * np.sin(3.0 * pw[..., 0]) * np.cos(2.5 * pw[..., 1]) + 0.2 * np.cos(2.0 * pw[..., 1] + 1.5 * pw[..., 2])
    return np.clip(tex, 0.0, 1.0).astype(np.float32), depth
