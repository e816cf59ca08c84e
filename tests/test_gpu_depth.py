"""GPU: depth / RGB-D image -> point cloud (PointCloud::CreateFromDepthImage,
CreateFromRGBDImage, pointcloud_factory.cu:286-376) and the KinFu pose estimation that
feeds those clouds to RegistrationICP level by level (kinfu.cpp:105-143), against the
oracle's restatement.  The kernel recomputes points instead of storing a structured cloud
and the compiler may contract a*b+c on the GPU, so coordinates are compared to 2 ulp-ish
(2e-6 relative); counts, validity and pixel order must be identical."""
import numpy as np
import pytest
import torch

from conftest import render_depth, small_pose
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

K = [525.0, 525.0, 319.5, 239.5]


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _close(a, b, tol=2e-6):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    fin = np.isfinite(b)
    assert (np.isfinite(a) == fin).all()
    assert (a[~fin] == b[~fin]).all()
    np.testing.assert_allclose(a[fin], b[fin], rtol=tol, atol=tol)


def _depth_with_defects(w=640, h=480, seed=3):
    d = render_depth(w, h, K, np.eye(4), holes=0.05, seed=seed)
    rng = np.random.default_rng(seed + 1)
    bad = rng.random(d.shape)
    d[bad < 0.01] = -1.0
    d[(bad > 0.01) & (bad < 0.02)] = np.nan
    d[(bad > 0.02) & (bad < 0.025)] = np.inf
    return d


@pytest.mark.parametrize("stride", [1, 2, 3, 7])
@pytest.mark.parametrize("with_extrinsic", [False, True])
def test_depth_image_float(eng, stride, with_extrinsic):
    d = _depth_with_defects()
    E = np.linalg.inv(small_pose(0.3, 0.5)) if with_extrinsic else None
    ref, _, _ = orc.create_from_depth(d, K, E, stride=stride)
    for dev in (False, True):
        got, n, c = eng.create_from_depth(torch.from_numpy(d).cuda() if dev else d, K, E, stride=stride)
        assert n is None and c is None
        _close(got.cpu().numpy() if dev else got, ref)
    assert 0 < len(ref) < (640 // stride) * (480 // stride)


def test_depth_image_u16_scale_and_trunc(eng):
    d = (np.clip(render_depth(320, 240, [262.5, 262.5, 159.5, 119.5], np.eye(4), holes=0.02), 0, 60) * 1000.0).astype(np.uint16)
    for scale, trunc in ((1000.0, 1000.0), (1000.0, 3.0), (999.7, 2.9), (500.0, 4.5)):
        ref, _, _ = orc.create_from_depth(d, [262.5, 262.5, 159.5, 119.5], None, depth_scale=scale, depth_trunc=trunc)
        got, _, _ = eng.create_from_depth(torch.from_numpy(d.view(np.int16)).cuda().view(torch.uint16),
                                          [262.5, 262.5, 159.5, 119.5], None, depth_scale=scale, depth_trunc=trunc)
        _close(got.cpu().numpy(), ref)
        got_h, _, _ = eng.create_from_depth(d, [262.5, 262.5, 159.5, 119.5], None, depth_scale=scale, depth_trunc=trunc)
        _close(got_h, ref)
    # the reference holds depth_trunc as an int: 2.9 behaves as 2
    a, _, _ = eng.create_from_depth(d, [262.5, 262.5, 159.5, 119.5], None, depth_trunc=2.9)
    b, _, _ = eng.create_from_depth(d, [262.5, 262.5, 159.5, 119.5], None, depth_trunc=2.0)
    assert len(a) == len(b) and len(a) > 0


@pytest.mark.parametrize("valid_only", [True, False])
@pytest.mark.parametrize("color_kind", ["u8", "f32", None])
def test_rgbd_image_colors_normals_cutoff(eng, valid_only, color_kind):
    d = _depth_with_defects(seed=9)
    rng = np.random.default_rng(5)
    col = None
    if color_kind == "u8":
        col = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    elif color_kind == "f32":
        col = rng.random((480, 640), dtype=np.float32)
    E = np.linalg.inv(small_pose(0.1, 0.2))
    kw = dict(color=col, depth_cutoff=3.0, rgbd=True, compute_normals=True, valid_only=valid_only)
    rp, rn, rc = orc.create_from_depth(d, K, E, **kw)
    gp, gn, gc = eng.create_from_depth(torch.from_numpy(d).cuda(), K, E,
                                       **dict(kw, color=None if col is None else torch.from_numpy(col).cuda()))
    _close(gp.cpu().numpy(), rp)
    if col is None:
        assert gc is None
    else:
        _close(gc.cpu().numpy(), rc, 1e-7)
    # normals: unit vectors from differences of nearby points -- compare by angle; where the
    # cross product is nearly degenerate the direction amplifies the 1e-7 coordinate noise
    gn = gn.cpu().numpy()
    assert gn.shape == rn.shape
    zero_r, zero_g = (np.abs(rn).sum(1) == 0), (np.abs(gn).sum(1) == 0)
    assert (zero_r == zero_g).mean() > 0.9999
    both = ~zero_r & ~zero_g
    cosang = (gn[both] * rn[both]).sum(1)
    assert (cosang > 1 - 1e-6).mean() > 0.999
    assert (gn[both][:, 2] <= 0).all()
    if valid_only:
        assert np.isfinite(gp.cpu().numpy()).all() and len(rp) < 640 * 480
        assert (gp.cpu().numpy()[:, 2] < 3.0 + 1.0).all()
    else:
        assert len(rp) == 640 * 480


def test_last_row_and_column_conventions(eng):
    """documented handling of the reference's loose bounds test (pointcloud_factory.cu:173)"""
    d = np.full((6, 8), 2.0, np.float32)
    d[2, 3] = 0.0
    _, n, _ = eng.create_from_depth(d, [4.0, 4.0, 3.5, 2.5], rgbd=True, compute_normals=True, valid_only=False)
    _, rn, _ = orc.create_from_depth(d, [4.0, 4.0, 3.5, 2.5], rgbd=True, compute_normals=True, valid_only=False)
    np.testing.assert_allclose(n, rn, atol=1e-6)
    n = n.reshape(6, 8, 3)
    assert (n[0] == 0).all() and (n[:, 0] == 0).all()          # first row / column: zero normal
    assert (n[1:5, 1:7, 2] < 0).all()                           # interior: facing the camera
    assert np.abs(n[5, 1:7]).sum() > 0                          # last row: lower neighbour taken as zero


def test_argument_errors(eng):
    from cupoch_amd import MiIcpError
    d = np.ones((4, 4), np.float32)
    with pytest.raises(MiIcpError):
        eng.create_from_depth(d, K, stride=0)
    with pytest.raises(MiIcpError):
        eng.create_from_depth(d.astype(np.uint16), K, rgbd=True)
    with pytest.raises(MiIcpError):
        eng.create_from_depth(d, K, compute_normals=True)              # normals belong to the RGB-D form
    with pytest.raises(MiIcpError):
        eng.create_from_depth(d, K, np.zeros((4, 4), np.float32))       # singular extrinsic
    p, _, _ = eng.create_from_depth(np.zeros((4, 4), np.float32), K)
    assert len(p) == 0
    p, _, _ = eng.create_from_depth(np.zeros((0, 0), np.float32), K)
    assert len(p) == 0


def test_python_factories_and_pyramid_intrinsics():
    from cupoch_amd import camera, geometry
    intr = camera.PinholeCameraIntrinsic(640, 480, *K)
    for level in range(4):
        w, h, fx, fy, cx, cy = orc.pyramid_level_intrinsic(640, 480, *K, level)
        lv = intr.create_pyramid_level(level)
        assert (lv.width, lv.height) == (w, h)
        np.testing.assert_array_equal(np.float32(lv.as4()), np.float32([fx, fy, cx, cy]))
    d = render_depth(640, 480, K, np.eye(4), holes=0.03)
    pc = geometry.PointCloud.create_from_depth_image(geometry.Image(d), intr, np.eye(4, dtype=np.float32), 1000.0, 1000.0, 2)
    ref, _, _ = orc.create_from_depth(d, K, stride=2)
    _close(pc.points.cpu(), ref)
    assert not pc.has_normals()
    col = np.random.default_rng(1).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    pc = geometry.PointCloud.create_from_rgbd_image(geometry.RGBDImage(col, torch.from_numpy(d)), intr,
                                                    compute_normals=True, depth_cutoff=3.5)
    rp, rn, rc = orc.create_from_depth(d, K, color=col, rgbd=True, compute_normals=True, depth_cutoff=3.5)
    _close(pc.points.cpu(), rp)
    _close(pc.colors.cpu(), rc, 1e-7)
    assert pc.has_normals() and pc.has_colors()
    bad = geometry.PointCloud.create_from_depth_image(np.zeros((4, 4, 3), np.uint8), intr)
    assert bad.is_empty()


def _pyramids(levels, pose_b, scale):
    """frame A at the identity pose (the 'model'), frame B at pose_b (the new frame); every
    level rendered with its own intrinsics (the half-pixel convention of CreatePyramidLevel)"""
    from cupoch_amd import camera
    intr = camera.PinholeCameraIntrinsic(640, 480, *K)
    da, db = [], []
    for i in range(levels):
        lv = intr.create_pyramid_level(i)
        da.append(render_depth(lv.width, lv.height, lv.as4(), np.eye(4), scale=scale))
        db.append(render_depth(lv.width, lv.height, lv.as4(), pose_b, scale=scale))
    return intr, da, db


@pytest.mark.parametrize("colored", [False, True])
def test_kinfu_pose_estimation_matches_oracle_and_truth(colored):
    """Scene in centimetres: the reference's fp32 colour-gradient fit is ill-conditioned at
    metre scale (tests/test_gpu_colored.py), and a tracker test should not hinge on that."""
    from cupoch_amd import kinfu, registration
    levels, S = 3, 100.0
    pose_b = small_pose(0.02, 0.03 * S)
    intr, depth_a, depth_b = _pyramids(levels, pose_b, S)
    opt = kinfu.KinfuOption(num_pyramid_levels=levels, depth_cutoff=6.0 * S, distance_threshold=0.03 * S,
                            icp_iterations=(10, 10, 10),
                            tf_type=(registration.TransformationEstimationType.ColoredICP if colored
                                     else registration.TransformationEstimationType.PointToPlane))
    cols_a = cols_b = None
    if colored:   # a smooth world-space texture, so both frames see consistent colours
        cols_a, cols_b = [], []
        for i in range(levels):
            for dd, pose, dst in ((depth_a[i], np.eye(4), cols_a), (depth_b[i], pose_b, cols_b)):
                k = intr.create_pyramid_level(i).as4()
                p, _, _ = orc.create_from_depth(dd, k, np.linalg.inv(pose), rgbd=True, valid_only=False)
                p = np.where(np.isfinite(p), p, 0.0) / S
                v = 0.5 + 0.25 * np.sin(3.0 * p[:, 0]) + 0.25 * np.cos(2.0 * p[:, 1] + p[:, 2])
                dst.append(v.astype(np.float32).reshape(dd.shape))
    model = kinfu.point_cloud_pyramid(depth_a, intr, opt, cols_a)
    frame = kinfu.point_cloud_pyramid(depth_b, intr, opt, cols_b)
    assert [len(p.points) for p in model] == [int((d > 0).sum()) for d in depth_a]
    T, ok = kinfu.pose_estimation(opt, np.eye(4, dtype=np.float32), frame, model)
    assert ok

    def err(A, B):
        D = np.asarray(A, np.float64) - np.asarray(B, np.float64)
        D[:3, 3] /= S
        return np.linalg.norm(D)
    # truth: the new frame's camera -> world pose (its points are in its camera frame);
    # a threshold as wide as the occlusion shadows (0.1 m) would bias this to 7e-3
    assert err(T, pose_b) < 2e-3
    as_dict = lambda pcs: [dict(points=p.points.cpu(), normals=p.normals.cpu(),
                                colors=(p.colors.cpu() if p.has_colors() else None)) for p in pcs]
    T_ref = orc.kinfu_pose_estimation(np.eye(4, dtype=np.float32), as_dict(frame), as_dict(model),
                                      distance_threshold=0.03 * S, icp_iterations=(10, 10, 10), colored=colored)
    assert err(T, T_ref) < 1e-4
