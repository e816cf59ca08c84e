"""GPU: real surface data, and BASELINE.json's full sizes through size-independent
properties (the oracle-checked counterparts at these sizes: test_gpu_baseline_configs.py)."""
import numpy as np
import pytest
import torch

from conftest import make_pair
from oracle import oracle as orc
from test_io_and_real_data import fragment

pytestmark = pytest.mark.gpu
PT2PL = 2


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_real_scan_matches_oracle(eng):
    pts, nrm = fragment()
    ang = 0.03
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
    T[:3, 3] = [0.004, -0.003, 0.002]
    src = orc.transform_points(np.linalg.inv(T).astype(np.float32), pts)
    eng.set_target(pts, nrm)
    eng.set_source(src)
    idx, d2, st = eng.search_radius_1nn(0.02)
    cnt, oi, od = orc.search_radius(pts, src, 0.02, 1)
    assert st[0] == cnt
    hit = oi[:, 0] >= 0
    np.testing.assert_array_equal(d2[hit], od[hit, 0])                 # surfaces: bit-exact distances
    assert (idx[hit] != oi[hit, 0]).sum() <= 2                          # ties only
    res = eng.registration_icp(PT2PL, 0.02, None, 1e-6, 1e-6, 30, -1.0)
    ref = orc.registration_icp(src, pts, 0.02, est=orc.EST_PT2PL, tgt_nrm=nrm, det_thresh=-1.0)
    Tg = np.array(res.transformation, np.float32).reshape(4, 4).T
    assert np.linalg.norm(Tg - ref.transformation) <= 1e-5
    assert res.fitness == pytest.approx(ref.fitness, abs=1e-4)
    # normals from the scan itself: EstimateNormals(KNN 30) agrees with the oracle's restatement
    got = eng.estimate_normals_knn(cuda(pts), 30).cpu().numpy()
    want = orc.estimate_normals_knn(pts, 30)
    dots = np.abs((got * want).sum(1))
    assert (dots > 1 - 1e-3).mean() > 0.99
    # and is a sane surface normal field: mostly aligned with the normals shipped with the scan
    assert (np.abs((got * nrm).sum(1)) > 0.9).mean() > 0.8


def test_ten_million_points_properties(eng):
    """BASELINE config 3 at full size: ICP recovers the ground-truth transform, the search
    is idempotent under seeding, correspondences are a valid ascending injection-free set."""
    n = 10_000_000
    d = make_pair(n, seed=42)
    eng.set_target(cuda(d["tgt"]), cuda(d["tgt_nrm"]))
    eng.set_source(cuda(d["src"]))
    # (1) search under the ground truth: every point finds its own pre-image at ~zero distance
    idx, d2, st = eng.search_radius_1nn(d["max_dist"], d["T_gt"])
    assert st[0] == n and (idx >= 0).all()
    assert d2.max() < (1e-5) ** 2
    perm = np.random.Generator(np.random.PCG64(44)).permutation(n)     # make_pair's source permutation
    assert (idx == perm).mean() > 0.999999
    # (2) a checksum of checksums: sum d2 reported by the reduction == sum of the per-point output
    np.testing.assert_allclose(st[1], d2.astype(np.float64).sum(), rtol=1e-9)
    # (3) full ICP from identity recovers T_gt; seeded iterations agree with an unseeded evaluation
    res = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, 20, -1.0)
    T = np.array(res.transformation, np.float32).reshape(4, 4).T
    assert np.linalg.norm(T - d["T_gt"]) < 1e-6 and res.fitness == 1.0
    eng.drop_seeds()                                                   # the next search starts top-down ...
    ev = eng.evaluate_registration(d["max_dist"], T)                   # ... a fresh, UNSEEDED search under T
    assert ev.fitness == res.fitness and ev.inlier_rmse == pytest.approx(res.inlier_rmse, rel=1e-3, abs=1e-9)
    cor = eng.get_correspondences()
    assert len(cor) == n and (np.diff(cor[:, 0]) > 0).all() and (cor[:, 0] == np.arange(n)).all()
    assert len(np.unique(cor[:, 1])) == n                              # a bijection on this data


def test_voxel_downsample_properties_at_scale(eng):
    n = 10_000_000
    rng = np.random.default_rng(1)
    pts = rng.random((n, 3), dtype=np.float32)
    v, _, _ = eng.voxel_downsample(cuda(pts), 0.02)
    v = v.cpu().numpy()
    origin = pts.min(0) - np.float32(0.02) * np.float32(0.5)
    def keys(a):
        c = np.floor((a - origin) / np.float32(0.02)).astype(np.int64)
        return (c[:, 0] * 4096 + c[:, 1]) * 4096 + c[:, 2]
    kin = np.unique(keys(pts))
    assert len(v) == len(kin)                          # exactly one output per occupied voxel
    kout = keys(v)
    assert (np.diff(kout) > 0).all()                   # lexicographic (x, y, z) voxel order
    assert (kout == kin).mean() > 0.9999               # a voxel mean stays inside its voxel (up to fp32 edges)
    np.testing.assert_allclose(v.mean(0), pts.mean(0), atol=2e-3)
