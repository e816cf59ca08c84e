"""PCD reader + the real-scan fixture on the CPU side (oracle only)."""
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as orc


def fragment():
    d = np.load(os.path.join(ROOT, "tests", "golden", "fragment_every3rd.npz"))
    nrm = d["normals"].astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return np.ascontiguousarray(d["points"]), nrm


def test_pcd_reader_roundtrip(tmp_path):
    from cupoch_amd.io import read_pcd_arrays
    rng = np.random.default_rng(0)
    pts = rng.random((50, 3), dtype=np.float32)
    nrm = rng.random((50, 3), dtype=np.float32)
    rgb = rng.integers(0, 256, (50, 3), dtype=np.uint32)
    packed = ((rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2]).astype(np.uint32).view(np.float32)
    rec = np.zeros(50, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgb", "<f4"),
                              ("normal_x", "<f4"), ("normal_y", "<f4"), ("normal_z", "<f4")])
    for i, k in enumerate("xyz"):
        rec[k] = pts[:, i]
        rec["normal_" + k] = nrm[:, i]
    rec["rgb"] = packed
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb normal_x normal_y normal_z\n"
           "SIZE 4 4 4 4 4 4 4\nTYPE F F F F F F F\nCOUNT 1 1 1 1 1 1 1\nWIDTH 50\nHEIGHT 1\n"
           "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS 50\nDATA binary\n")
    p = tmp_path / "t.pcd"
    p.write_bytes(hdr.encode() + rec.tobytes())
    a = read_pcd_arrays(str(p))
    np.testing.assert_array_equal(a["points"], pts)
    np.testing.assert_array_equal(a["normals"], nrm)
    np.testing.assert_allclose(a["colors"], rgb / 255.0, atol=1e-6)
    asc = tmp_path / "a.pcd"
    asc.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\n"
                   "DATA ascii\n1 2 3\n4 5 6\n")
    a = read_pcd_arrays(str(asc))
    np.testing.assert_array_equal(a["points"], [[1, 2, 3], [4, 5, 6]])
    assert a["normals"] is None and a["colors"] is None


def test_oracle_icp_on_real_scan():
    """The example flow of examples/python/basic/icp_registration.py (threshold 0.02,
    point-to-plane) on the reference's sample scan against a moved copy of itself."""
    pts, nrm = fragment()
    ang = 0.03
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
    T[:3, 3] = [0.004, -0.003, 0.002]
    src = orc.transform_points(np.linalg.inv(T).astype(np.float32), pts)
    res = orc.registration_icp(src, pts, 0.02, est=orc.EST_PT2PL, tgt_nrm=nrm, det_thresh=-1.0)
    assert res.fitness > 0.99
    assert np.linalg.norm(res.transformation - T) < 2e-4
