"""GPU parity at BASELINE.json's NAMED configurations, at their sizes.

config 3: 10M-vs-10M point-to-plane -- engine indices / d2 BIT-EXACT against the oracle's
          kd-tree (full 10M target) on a 1M-query sample, unseeded, seeded under a different
          transform, and after the loop's match-order re-sort; the 6x6 system over all 10M
          correspondences against the oracle's O(N) accumulation (1e-9).
config 2: 1M -> VoxelDownSample(0.02) both -> point-to-plane r = 0.04, end to end against
          oracle voxel_downsample + oracle registration_icp.
config 5: GeneralizedICP 5M-vs-5M -- covariances from normals, system (2e-5: acosf/cosf ulps),
          final transform after a fixed number of iterations (1e-5 Frobenius).
The oracle runs on this box's host cores (OpenMP); every case finishes in tens of seconds.
"""
import numpy as np
import pytest
import torch

from conftest import make_pair
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
P2P, PT2PL, GICP = 1, 2, 5


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def cuda(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def fraction_of(T_gt, f, spacing):
    """a rigid transform a fraction f of the way to T_gt (same axis, angle and shift scaled)"""
    ang = 0.2 * spacing * f
    ax = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    T[:3, 3] = T_gt[:3, 3].astype(np.float64) * f
    return T.astype(np.float32)


def assert_sample_bit_exact(idx, d2, sample, tree, src, tgt, T, radius):
    """engine result (original order) on `sample` == the oracle's, bit for bit; an index may
    differ only between two target points at exactly the same fp32 distance"""
    q = src[sample] if T is None else orc.transform_points(T, src[sample])
    _, oi, od = tree.search_radius(q, radius, 1)
    oi, od = oi[:, 0], od[:, 0]
    gi, gd = idx[sample], d2[sample]
    assert np.array_equal(gi < 0, oi < 0), "hit/miss pattern differs"
    hit = oi >= 0
    assert np.array_equal(gd[hit], od[hit]), "d2 not bit-exact"
    assert np.isinf(gd[~hit]).all()
    diff = np.flatnonzero(gi != oi)
    if len(diff):
        dd = q[diff] - tgt[gi[diff]]
        alt = (dd[:, 2] * dd[:, 2] + (dd[:, 1] * dd[:, 1] + dd[:, 0] * dd[:, 0])).astype(np.float32)
        assert np.array_equal(alt, od[diff]), "index mismatch that is not an exact tie"
    return int(hit.sum()), len(diff)


def test_config3_ten_million_nn_bit_exact_and_system(eng):
    n = 10_000_000
    d = make_pair(n, seed=42)
    src, tgt, nrm, r, s = d["src"], d["tgt"], d["tgt_nrm"], d["max_dist"], d["spacing"]
    tree = orc.Tree(tgt)
    eng.set_target(cuda(tgt), cuda(nrm))
    eng.set_source(cuda(src))
    sample = np.sort(np.random.default_rng(7).choice(n, 1_000_000, replace=False))

    # (a) identity, unseeded first pass
    idx, d2, st = eng.search_radius_1nn(r)
    hits, ties = assert_sample_bit_exact(idx, d2, sample, tree, src, tgt, None, r)
    assert hits > 900_000
    # (b) mid-trajectory transform, SEEDED by (a)
    T_mid = fraction_of(d["T_gt"], 0.55, s)
    idx, d2, _ = eng.search_radius_1nn(r, T_mid)
    assert_sample_bit_exact(idx, d2, sample, tree, src, tgt, T_mid, r)
    # (c) the same transform from scratch: identical output, element for element
    eng.drop_seeds()
    idx_u, d2_u, _ = eng.search_radius_1nn(r, T_mid)
    assert np.array_equal(d2, d2_u)
    assert (idx != idx_u).sum() == 0 or np.array_equal(d2[idx != idx_u], d2_u[idx != idx_u])
    # (d) the loop's state: first pass + match-order re-sort under T_mid, then seeded searches
    # under a DIFFERENT transform (large step), a tiny step from it, and the ground truth
    eng.icp_begin(PT2PL, r, T_mid, -1.0)
    for T in (fraction_of(d["T_gt"], 0.9, s), fraction_of(d["T_gt"], 0.9001, s), d["T_gt"]):
        idx, d2, st = eng.search_radius_1nn(r, T)
        assert_sample_bit_exact(idx, d2, sample, tree, src, tgt, T, r)
    # under T_gt every source point is its pre-image: all 10M matched
    assert st[0] == n
    # (e) the point-to-plane system over ALL 10M correspondences against the oracle's O(N) sum
    got = eng.compute_system(PT2PL, d["T_gt"])
    cor = eng.get_correspondences()
    assert len(cor) == n and (cor[:, 0] == np.arange(n)).all()
    src_t = orc.transform_points(d["T_gt"], src)
    ref = orc.compute_system(PT2PL, src_t, tgt, cor, None, nrm)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
    # (f) and at the mid-trajectory transform (misses + non-trivial residuals)
    idx, d2, st = eng.search_radius_1nn(r, T_mid)
    got = eng.compute_system(PT2PL, T_mid)
    cor = eng.get_correspondences()
    keep = idx >= 0
    np.testing.assert_array_equal(cor, np.stack([np.flatnonzero(keep), idx[keep]], 1).astype(np.int32))
    ref = orc.compute_system(PT2PL, orc.transform_points(T_mid, src), tgt, cor, None, nrm)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
    tree.close()


def test_config2_million_voxel_point_to_plane_end_to_end(eng):
    n = 1_000_000
    d = make_pair(n, seed=42)
    voxel, r = 0.02, 0.04
    vt, vtn, _ = eng.voxel_downsample(cuda(d["tgt"]), voxel, cuda(d["tgt_nrm"]))
    vs, _, _ = eng.voxel_downsample(cuda(d["src"]), voxel)
    vt, vtn, vs = vt.cpu().numpy(), vtn.cpu().numpy(), vs.cpu().numpy()
    ot, otn, _ = orc.voxel_downsample(d["tgt"], voxel, d["tgt_nrm"])
    os_, _, _ = orc.voxel_downsample(d["src"], voxel)
    # output count and (lexicographic voxel) order identical; means to fp32 rounding of an fp64 mean
    assert len(vt) == len(ot) and len(vs) == len(os_) and 100_000 < len(vt) < 140_000
    np.testing.assert_allclose(vt, ot, atol=2e-7)
    np.testing.assert_allclose(vs, os_, atol=2e-7)
    np.testing.assert_allclose(vtn, otn, atol=2e-6)
    # ICP on the engine's own downsampled clouds vs the oracle on the oracle's
    eng.set_target(cuda(vt), cuda(vtn))
    eng.set_source(cuda(vs))
    res = eng.registration_icp(PT2PL, r, None, 1e-6, 1e-6, 30, -1.0)
    T = np.array(res.transformation, np.float32).reshape(4, 4).T
    ref = orc.registration_icp(os_, ot, r, est=orc.EST_PT2PL, tgt_nrm=otn, det_thresh=-1.0)
    assert np.linalg.norm(T - ref.transformation) <= 1e-5
    assert res.iterations == ref.iterations
    assert res.fitness == pytest.approx(ref.fitness, abs=1e-5)
    assert res.inlier_rmse == pytest.approx(ref.inlier_rmse, rel=1e-4)
    # same inputs on both sides: the correspondence sets are identical
    ref2 = orc.registration_icp(vs, vt, r, est=orc.EST_PT2PL, tgt_nrm=vtn, det_thresh=-1.0)
    assert np.linalg.norm(T - ref2.transformation) <= 1e-5 and res.iterations == ref2.iterations
    cor = eng.get_correspondences()
    a, b = set(map(tuple, cor.tolist())), set(map(tuple, ref2.correspondence_set.tolist()))
    assert len(a ^ b) <= 2, "correspondence sets differ beyond last-ulp flips: %d" % len(a ^ b)


def test_config5_generalized_icp_five_million(eng):
    n = 5_000_000
    d = make_pair(n, seed=42)
    src, tgt, r = d["src"], d["tgt"], d["max_dist"]
    tcov_d = eng.covariances_from_normals(cuda(d["tgt_nrm"]), 1e-3)
    scov_d = eng.covariances_from_normals(cuda(d["src_nrm"]), 1e-3)
    tcov, scov = tcov_d.cpu().numpy(), scov_d.cpu().numpy()
    # InitializePointCloudForGeneralizedICP's covariances: a 200k sample against the oracle
    ref_cov = orc.covariances_from_normals(d["tgt_nrm"][:200_000], 1e-3)
    np.testing.assert_allclose(tcov[:200_000].reshape(-1, 3, 3), ref_cov, atol=2e-6)
    eng.set_target(cuda(tgt), None, tcov_d)
    eng.set_source(cuda(src), None, scov_d)
    # system under a mid-trajectory transform over all 5M points
    T_mid = fraction_of(d["T_gt"], 0.6, d["spacing"])
    idx, _, _ = eng.search_radius_1nn(r, T_mid)
    got = eng.compute_system(GICP, T_mid)
    cor = eng.get_correspondences()
    assert len(cor) > 0.9 * n
    src_t = orc.transform_points(T_mid, src)
    cov_t = orc.rotate_covariances(T_mid, scov.reshape(-1, 3, 3))
    ref = orc.compute_system(GICP, src_t, tgt, cor, None, None, cov_t, tcov.reshape(-1, 3, 3))
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    # the registration itself, fixed iteration count (6 oracle passes over 5M points)
    res = eng.registration_icp(GICP, r, None, 0.0, 0.0, 5, -1.0)
    T = np.array(res.transformation, np.float32).reshape(4, 4).T
    ref = orc.registration_icp(src, tgt, r, est=orc.EST_GICP, det_thresh=-1.0, relative_fitness=0.0,
                               relative_rmse=0.0, max_iteration=5, src_cov=scov.reshape(-1, 3, 3),
                               tgt_cov=tcov.reshape(-1, 3, 3))
    assert res.iterations == 5 and ref.iterations == 5
    assert np.linalg.norm(T - ref.transformation) <= 1e-5
    assert res.fitness == pytest.approx(ref.fitness, abs=1e-6)
    assert np.linalg.norm(T - d["T_gt"]) <= 1e-4
