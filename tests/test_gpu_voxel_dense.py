"""VoxelDownSample's dense-grid path (csrc/voxel_dense.h: one 2048-way stable partition, then a workgroup per bucket)
against the CPU oracle (oracle/icp_oracle.c oracle_voxel_downsample, geometry/down_sample.cu:64-90,170-273) and against
the general path (radix passes with the payload, geometry_kernels.h) on the same inputs.

The dense path adds a voxel's points in INPUT order in fp64 -- the oracle's order -- so its means are compared bit for
bit; the switch MI_ICP_NO_DENSE_VOXEL is read at every call, which is what lets one process run both paths."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def cuda(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def general(eng, *args):
    os.environ["MI_ICP_NO_DENSE_VOXEL"] = "1"
    try:
        return eng.voxel_downsample(*args)
    finally:
        os.environ.pop("MI_ICP_NO_DENSE_VOXEL", None)


def dense(eng, *args):
    os.environ.pop("MI_ICP_NO_DENSE_VOXEL", None)
    return eng.voxel_downsample(*args)


def took_dense_path(eng):
    """the path of the last call, from the library's debug word (1: dense, 0: general)"""
    return eng._L.mi_icp_debug_last_voxel_path(eng._ctx) == 1


# (n, voxel, extent per axis, offset): the plan (voxel_dense.h vx_make_plan) takes keys of 14 ... 22 bits with at least 256
# points per bucket; these cover 2048 ... 16 buckets, 2048 ... 64 voxels per bucket (one or two per thread of the
# finishing kernel), buckets of several LDS chunks (6M points in 512 buckets), a slab (the key's bits all in x and y),
# a cloud far from the origin and one that straddles it
CASES = [
    (800_000, 0.01, (1.0, 1.0, 1.0), 0.0),
    (200_000, 0.02, (1.0, 1.0, 1.0), 0.0),
    (1_000_000, 0.02, (1.0, 1.0, 1.0), -0.5),
    (400_000, 0.05, (1.0, 1.0, 1.0), 0.0),
    (3_000_000, 0.01, (1.0, 1.0, 1.0), 100.0),
    (6_000_000, 0.05, (1.0, 1.0, 1.0), 0.0),
    (500_000, 0.03, (1.0, 0.3, 2.0), -0.25),
    (700_000, 0.011, (1.0, 1.0, 1.0), 0.0),
    (300_000, 0.02, (4.0, 4.0, 0.03), 0.0),
    (131_072, 0.04, (1.0, 1.0, 1.0), 0.0),
    (2_500_000, 0.01, (1.3, 1.0, 1.0), 0.0),                     # a 22-bit key: 2048 buckets of 2048 voxels
    (10_000_000, 0.01, (1.0, 1.0, 1.0), 0.0),                    # the bench's shape: 1024 buckets of ~9.8k points, two LDS chunks each, two voxels per thread
]


@pytest.mark.parametrize("n,voxel,ext,off", CASES)
def test_dense_path_equals_the_oracle_bit_for_bit(eng, n, voxel, ext, off):
    rng = np.random.default_rng(n)
    pts = (rng.random((n, 3), dtype=np.float32) * np.asarray(ext, np.float32) + np.float32(off)).astype(np.float32)
    nrm = rng.standard_normal((n, 3)).astype(np.float32)
    col = rng.random((n, 3), dtype=np.float32)
    p, nn, c = dense(eng, cuda(pts), voxel, cuda(nrm), cuda(col))
    assert took_dense_path(eng), "the case was written for the dense path"
    rp, rn, rc = orc.voxel_downsample(pts, voxel, nrm, col)
    assert len(p) == len(rp)                                     # same voxels, same (lexicographic) order
    np.testing.assert_array_equal(p.cpu().numpy(), rp)
    np.testing.assert_array_equal(c.cpu().numpy(), rc)
    np.testing.assert_array_equal(nn.cpu().numpy(), rn)
    gp, gn, gc = general(eng, cuda(pts), voxel, cuda(nrm), cuda(col))
    assert not took_dense_path(eng)
    assert len(gp) == len(rp)
    np.testing.assert_allclose(gp.cpu().numpy(), rp, atol=1e-6)
    # every subset of the arrays (the kernels are instantiated per subset), device and host arrays
    p1, n1, c1 = dense(eng, cuda(pts), voxel)
    assert n1 is None and c1 is None
    np.testing.assert_array_equal(p1.cpu().numpy(), rp)
    p2, n2, c2 = dense(eng, cuda(pts), voxel, None, cuda(col))
    assert n2 is None
    np.testing.assert_array_equal(p2.cpu().numpy(), rp)
    np.testing.assert_array_equal(c2.cpu().numpy(), rc)
    if n <= 1_000_000:
        p3, n3, c3 = dense(eng, pts, voxel, nrm, None)
        assert c3 is None and took_dense_path(eng)
        np.testing.assert_array_equal(np.asarray(p3), rp)
        np.testing.assert_array_equal(np.asarray(n3), rn)


def test_sums_are_made_in_input_order(eng):
    """A payload whose fp64 sum depends on the order of addition: every voxel holds, among ordinary colours, the values
    1e30, 1 and -1e30 at random places -- (1e30 + 1) - 1e30 = 0 but (1e30 - 1e30) + 1 = 1 in fp64.  Any element out of
    input order inside a bucket (the partition's ranks: LDS adds served in lane order, voxel_dense.h "Ranks") or inside a
    voxel's run (the finishing kernel's counting sort) changes some voxel's mean by ~1/count."""
    rng = np.random.default_rng(11)
    n, voxel = 1_500_000, 0.02
    pts = rng.random((n, 3), dtype=np.float32)
    col = rng.random((n, 3), dtype=np.float32)
    pick = rng.permutation(n)[: n // 4]
    col[pick[0::3], 0] = 1e30
    col[pick[1::3], 0] = -1e30
    col[pick[2::3], 1] = 3e29
    col[pick[0::3], 2] = -7e29
    for _ in range(3):
        p, _, c = dense(eng, cuda(pts), voxel, None, cuda(col))
        assert took_dense_path(eng)
        rp, _, rc = orc.voxel_downsample(pts, voxel, None, col)
        np.testing.assert_array_equal(p.cpu().numpy(), rp)
        np.testing.assert_array_equal(c.cpu().numpy(), rc)


def test_a_crowded_cloud_is_left_to_the_general_path(eng):
    """600k points in one corner of a grid that 2000 scattered points stretch: one bucket would hold nearly all of them,
    and the finishing kernel gives a bucket to ONE workgroup -- vx_colscan flags it, nothing is written, the general
    path takes the call."""
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.random((600_000, 3), dtype=np.float32) * np.float32(0.02),
                          rng.random((2_000, 3), dtype=np.float32)]).astype(np.float32)
    col = rng.random((len(pts), 3), dtype=np.float32)
    p, _, c = dense(eng, cuda(pts), 0.01, None, cuda(col))
    assert not took_dense_path(eng)
    rp, _, rc = orc.voxel_downsample(pts, 0.01, None, col)
    assert len(p) == len(rp)
    np.testing.assert_allclose(p.cpu().numpy(), rp, atol=1e-6)
    np.testing.assert_allclose(c.cpu().numpy(), rc, atol=2e-6)


def test_grids_outside_the_plan_take_the_general_path(eng):
    rng = np.random.default_rng(2)
    pts = rng.random((400_000, 3), dtype=np.float32)
    for voxel, why in ((0.001, "30 key bits"), (0.2, "9 key bits")):
        p, _, _ = dense(eng, cuda(pts), voxel)
        assert not took_dense_path(eng), why
        rp, _, _ = orc.voxel_downsample(pts, voxel)
        assert len(p) == len(rp)
        np.testing.assert_allclose(p.cpu().numpy(), rp, atol=1e-6)
    small = pts[:100_000]                                        # fewer than 2^17 points: not even started
    p, _, _ = dense(eng, cuda(small), 0.02)
    assert not took_dense_path(eng)
    assert len(p) == len(orc.voxel_downsample(small, 0.02)[0])


def test_same_bits_from_run_to_run(eng):
    rng = np.random.default_rng(1)
    pts = cuda(rng.random((2_000_000, 3), dtype=np.float32))
    nrm = cuda(rng.standard_normal((2_000_000, 3)).astype(np.float32))
    p0, n0, _ = dense(eng, pts, 0.01, nrm)
    p0, n0 = p0.clone(), n0.clone()
    for _ in range(5):
        p, nn, _ = dense(eng, pts, 0.01, nrm)
        assert torch.equal(p, p0) and torch.equal(nn, n0)


def test_points_exactly_on_cell_faces_and_non_finite_coordinates(eng):
    """Keys: the reciprocal estimate with the division as the fallback near an integer (voxel_dense.h "Keys") -- points ON
    the faces of the cells (multiples of the voxel size from the grid's origin, the quotient an exact integer or one ulp
    off it) must land where the division puts them; and a NaN coordinate must neither crash the path nor move any
    other voxel (its own cell index is whatever the conversion of NaN gives: not compared)."""
    rng = np.random.default_rng(3)
    n, voxel = 600_000, np.float32(0.0125)
    pts = rng.random((n, 3), dtype=np.float32)
    origin = pts.min(axis=0) - voxel * np.float32(0.5)
    k = rng.integers(0, 80, size=(n // 2, 3)).astype(np.float32)
    on_face = origin + k * voxel                                  # fp32 arithmetic: on a face up to rounding
    on_face = np.nextafter(on_face, on_face + rng.choice(np.float32([-1, 0, 1]), size=on_face.shape)).astype(np.float32)
    pts[: n // 2] = np.clip(on_face, pts.min(axis=0), pts.max(axis=0))
    p, _, _ = dense(eng, cuda(pts), float(voxel))
    assert took_dense_path(eng)
    rp, _, _ = orc.voxel_downsample(pts, float(voxel))
    assert len(p) == len(rp)
    np.testing.assert_array_equal(p.cpu().numpy(), rp)
    bad = pts.copy()
    bad[12345, 1] = np.nan
    pn, _, _ = dense(eng, cuda(bad), float(voxel))
    assert took_dense_path(eng)
    assert abs(len(pn) - len(rp)) <= 1


@pytest.mark.parametrize("n,voxel,off", [(600_000, 0.004, 0.0), (500_000, 0.0012, -3.0), (250_000, 0.002, 40.0), (1_500_000, 0.0031, 0.0)])
def test_large_clouds_on_fine_grids_take_the_wide_sort(eng, n, voxel, off):
    """Beyond the dense path's 22 key bits a cloud of >= 2^17 points is sorted whole by 11-bit digits with the dense path's
    partition kernels (mi_geometry.hip voxel_wide_sort: 24, 30, 27 and 27 key bits here -- three passes), the keys made
    once from the sorted points, and a thread per voxel walks its run in input order: the oracle's sums, bit for bit."""
    rng = np.random.default_rng(n)
    pts = (rng.random((n, 3), dtype=np.float32) + np.float32(off)).astype(np.float32)
    nrm = rng.standard_normal((n, 3)).astype(np.float32)
    col = rng.random((n, 3), dtype=np.float32)
    p, nn, c = dense(eng, cuda(pts), voxel, cuda(nrm), cuda(col))
    assert not took_dense_path(eng)
    rp, rn, rc = orc.voxel_downsample(pts, voxel, nrm, col)
    assert len(p) == len(rp) and len(rp) > n // 3                # (a fine grid: short runs)
    np.testing.assert_array_equal(p.cpu().numpy(), rp)
    np.testing.assert_array_equal(c.cpu().numpy(), rc)
    np.testing.assert_array_equal(nn.cpu().numpy(), rn)
    p1, _, _ = dense(eng, cuda(pts), voxel)
    np.testing.assert_array_equal(p1.cpu().numpy(), rp)
    p2, _, c2 = dense(eng, pts, voxel, None, col)                # host arrays
    np.testing.assert_array_equal(np.asarray(p2), rp)
    np.testing.assert_array_equal(np.asarray(c2), rc)

