"""GPU: RGB-D odometry (odometry::ComputeRGBDOdometry, odometry/odometry.cu) through the C ABI
against the oracle's restatement (whose Jacobian functors are pinned by the reference's golden
vectors, tests/test_oracle_golden.py).  The kernels follow the reference's fp32 arithmetic order
(no contraction), the sums are fp64 in both: transformations agree to ~1e-6; the tolerance is 1e-4
because a correspondence decision (rounding to a pixel, the depth-difference test) can flip on the
last ulp of the running transformation."""
import numpy as np
import pytest
import torch

from conftest import render_rgbd, small_pose
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

K = [262.5, 262.5, 159.5, 119.5]


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def frames():
    pose_b = small_pose(0.02, 0.03)
    ca, da = render_rgbd(320, 240, K, np.eye(4), holes=0.02, seed=1)
    cb, db = render_rgbd(320, 240, K, pose_b, holes=0.02, seed=2)
    return pose_b, ca, da, cb, db


@pytest.mark.parametrize("jac", [0, 1])
@pytest.mark.parametrize("on_device", [False, True])
def test_odometry_matches_oracle_and_truth(eng, frames, jac, on_device):
    pose_b, ca, da, cb, db = frames
    conv = (lambda x: torch.from_numpy(x).cuda()) if on_device else (lambda x: x)
    ok, T, info = eng.compute_rgbd_odometry(conv(cb), conv(db), conv(ca), conv(da), K, None, jac, (20, 10, 5), 0.03, 0.0, 6.0)
    ok_r, T_r, info_r = orc.compute_rgbd_odometry(cb, db, ca, da, K, jacobian=jac, max_depth=6.0)
    assert ok and ok_r
    assert np.linalg.norm(T - T_r) < 1e-4, np.linalg.norm(T - T_r)
    np.testing.assert_allclose(info, info_r, rtol=2e-3)
    motion = np.linalg.norm(np.eye(4) - pose_b)
    assert np.linalg.norm(T - pose_b) < (0.1 if jac == 1 else 0.5) * motion


def test_single_level_and_init_and_options(eng, frames):
    pose_b, ca, da, cb, db = frames
    for kw in (dict(iterations=(7,), odo_init=None), dict(iterations=(3, 3), odo_init=pose_b),
               dict(iterations=(2, 2, 2, 2), odo_init=np.zeros((4, 4), np.float32)),     # isZero() -> identity
               dict(iterations=(4, 4), odo_init=None, max_depth_diff=0.07, min_depth=0.5, max_depth=3.5)):
        it = kw.pop("iterations")
        init = kw.pop("odo_init")
        mdd, mn, mx = kw.get("max_depth_diff", 0.03), kw.get("min_depth", 0.0), kw.get("max_depth", 6.0)
        ok, T, info = eng.compute_rgbd_odometry(cb, db, ca, da, K, init, 1, it, mdd, mn, mx)
        ok_r, T_r, info_r = orc.compute_rgbd_odometry(cb, db, ca, da, K, odo_init=init, jacobian=1, iterations=it,
                                                      max_depth_diff=mdd, min_depth=mn, max_depth=mx)
        assert ok == ok_r
        if not np.isfinite(T_r).all():      # the reference's "solution" of an empty system: NaN there, NaN here
            assert not np.isfinite(T).all()
            continue
        assert np.linalg.norm(T - T_r) < 1e-4
        np.testing.assert_allclose(info, info_r, rtol=2e-3)


def test_python_surface_and_errors(eng, frames):
    from cupoch_amd import camera, geometry, odometry, MiIcpError
    pose_b, ca, da, cb, db = frames
    intr = camera.PinholeCameraIntrinsic(320, 240, *K)
    src, tgt = geometry.RGBDImage(cb, db), geometry.RGBDImage(ca, da)
    ok, T, info = odometry.compute_rgbd_odometry(src, tgt, intr, np.eye(4, dtype=np.float32),
                                                 odometry.RGBDOdometryJacobianFromHybridTerm(),
                                                 odometry.OdometryOption(max_depth=6.0))
    _, T_r, _ = orc.compute_rgbd_odometry(cb, db, ca, da, K, jacobian=1, max_depth=6.0)
    assert ok and np.linalg.norm(T - T_r) < 1e-4 and info.shape == (6, 6)
    # a size mismatch is the reference's warning + failure tuple, not an exception
    bad = geometry.RGBDImage(ca[:100], da[:100])
    ok2, T2, info2 = odometry.compute_rgbd_odometry(src, bad, intr)
    assert not ok2 and np.array_equal(T2, np.eye(4, dtype=np.float32))
    with pytest.raises(MiIcpError):
        eng.compute_rgbd_odometry(cb, db, ca, da, K, None, 5)                      # unknown jacobian
    with pytest.raises(MiIcpError):
        eng.compute_rgbd_odometry(cb, db, ca, da, K, None, 1, iterations=(1,) * 9)  # too many levels


def test_weighted_odometry_matches_oracle(eng, frames):
    """ComputeWeightedRGBDOdometry: t-distribution weights, motion prior, the call's velocity as a twist"""
    pose_b, ca, da, cb, db = frames
    for kw in (dict(), dict(prev_twist=orc.matrix4_to_vector6(pose_b), inv_sigma_mat_diag=[500.0] * 6),
               dict(nu=3.0, sigma2_init=0.5, iterations=(6, 4))):
        it = kw.pop("iterations", (20, 10, 5))
        ok, T, tw, info = eng.compute_rgbd_odometry(cb, db, ca, da, K, None, 1, it, 0.03, 0.0, 6.0, True,
                                                    kw.get("prev_twist"), kw.get("nu", 5.0), kw.get("sigma2_init", 1.0),
                                                    kw.get("inv_sigma_mat_diag"))
        ok_r, T_r, tw_r, info_r = orc.compute_weighted_rgbd_odometry(cb, db, ca, da, K, iterations=it, max_depth=6.0, **kw)
        assert ok and ok_r
        assert np.linalg.norm(T - T_r) < 1e-4 and np.linalg.norm(tw - tw_r) < 1e-4
        np.testing.assert_allclose(info, info_r, rtol=2e-3)
        if it == (20, 10, 5):
            assert np.linalg.norm(T - pose_b) < 0.1 * np.linalg.norm(np.eye(4) - pose_b)
        np.testing.assert_allclose(tw, orc.matrix4_to_vector6(T), atol=1e-5)   # started from identity: velocity = result
    from cupoch_amd import camera, geometry, odometry
    ok, T, tw, info = odometry.compute_weighted_rgbd_odometry(geometry.RGBDImage(cb, db), geometry.RGBDImage(ca, da),
                                                             camera.PinholeCameraIntrinsic(320, 240, *K),
                                                             option=odometry.OdometryOption(max_depth=6.0))
    assert ok and tw.shape == (6,) and np.linalg.norm(T - pose_b) < 0.1 * np.linalg.norm(np.eye(4) - pose_b)
