"""GPU: the Python surface used the way cupoch's own examples use cupoch
(examples/python/basic/{icp_registration,gicp_registration,numpy_interop}.py): same
module / class / function names and defaults, results against the oracle."""
import numpy as np
import pytest
import torch

from conftest import make_pair
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def clouds(n=30000, seed=4, noise=0.0):
    import cupoch_amd as cph
    from cupoch_amd import geometry, registration, utility, io  # noqa: F401 -- attribute access below
    d = make_pair(n, seed=seed, noise=noise)
    source, target = cph.geometry.PointCloud(), cph.geometry.PointCloud()
    source.points = cph.utility.Vector3fVector(d["src"])          # numpy_interop.py idiom
    target.points = cph.utility.Vector3fVector(d["tgt"])
    target.normals = cph.utility.Vector3fVector(d["tgt_nrm"])
    source.normals = cph.utility.Vector3fVector(d["src_nrm"])
    return cph, d, source, target


def test_icp_registration_example_flow():
    cph, d, source, target = clouds()
    threshold = d["max_dist"]
    trans_init = np.eye(4, dtype=np.float32)
    ev = cph.registration.evaluate_registration(source, target, threshold, trans_init)
    oev = orc.evaluate_registration(d["src"], d["tgt"], threshold)
    assert abs(ev.fitness - oev.fitness) < 1e-6 and len(ev.correspondence_set) == len(oev.correspondence_set)
    assert np.array_equal(np.asarray(ev.correspondence_set), oev.correspondence_set)

    reg_p2p = cph.registration.registration_icp(
        source, target, threshold, trans_init, cph.registration.TransformationEstimationPointToPoint())
    o = orc.registration_icp(d["src"], d["tgt"], threshold, est=orc.EST_P2P)
    assert np.linalg.norm(reg_p2p.transformation - o.transformation) <= 1e-5
    assert "RegistrationResult" in repr(reg_p2p) and reg_p2p.transformation.shape == (4, 4)
    cs = np.asarray(reg_p2p.correspondence_set)
    assert cs.dtype == np.int32 and np.all(np.diff(cs[:, 0]) > 0)

    reg_p2l = cph.registration.registration_icp(
        source, target, threshold, trans_init, cph.registration.TransformationEstimationPointToPlane())
    o = orc.registration_icp(d["src"], d["tgt"], threshold, est=orc.EST_PT2PL, tgt_nrm=d["tgt_nrm"])
    assert np.linalg.norm(reg_p2l.transformation - o.transformation) <= 1e-5
    assert np.linalg.norm(reg_p2l.transformation - d["T_gt"]) < 1e-4

    reg_sym = cph.registration.registration_icp(
        source, target, threshold, trans_init, cph.registration.TransformationEstimationSymmetricMethod(),
        cph.registration.ICPConvergenceCriteria(max_iteration=20))
    o = orc.registration_icp(d["src"], d["tgt"], threshold, est=orc.EST_SYM, src_nrm=d["src_nrm"],
                             tgt_nrm=d["tgt_nrm"], max_iteration=20)
    assert np.linalg.norm(reg_sym.transformation - o.transformation) <= 1e-5

    reg_gicp = cph.registration.registration_generalized_icp(source, target, threshold, trans_init)
    assert np.linalg.norm(reg_gicp.transformation - d["T_gt"]) < 1e-3
    # inputs are never modified (registration.cu:147 deep-copies the source)
    assert np.array_equal(np.asarray(source.points.cpu()), d["src"])


class NumpyKabsch:
    """a user-defined estimator in pure numpy: the generic loop must call it every iteration"""
    calls = 0

    def get_transformation_estimation_type(self):
        from cupoch_amd.registration import TransformationEstimationType
        return TransformationEstimationType.Unspecified

    def compute_rmse(self, source, target, corres):
        return 0.0

    def compute_transformation(self, source, target, corres):
        NumpyKabsch.calls += 1
        cs = np.asarray(corres)
        if len(cs) == 0:
            return np.eye(4, dtype=np.float32)
        s = np.asarray(source.points.cpu(), np.float64)[cs[:, 0]]
        t = np.asarray(target.points.cpu(), np.float64)[cs[:, 1]]
        cs_, ct_ = s.mean(0), t.mean(0)
        U, _, Vt = np.linalg.svd((s - cs_).T @ (t - ct_))
        R = Vt.T @ np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))]) @ U.T
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, ct_ - R @ cs_
        return T.astype(np.float32)


def test_user_defined_estimators_go_through_the_generic_loop():
    cph, d, source, target = clouds(20000, seed=9)
    from cupoch_amd.registration import TransformationEstimation

    class Mine(NumpyKabsch, TransformationEstimation):
        pass

    NumpyKabsch.calls = 0
    res = cph.registration.registration_icp(source, target, d["max_dist"], np.eye(4, dtype=np.float32), Mine(),
                                            cph.registration.ICPConvergenceCriteria(max_iteration=12))
    assert NumpyKabsch.calls >= 2
    assert np.linalg.norm(res.transformation - d["T_gt"]) < 1e-4
    assert res.fitness > 0.999 and len(res.correspondence_set) > 19900

    class Wrapped(TransformationEstimation):      # goes through the engine itself: clouds get reloaded
        inner = cph.registration.TransformationEstimationPointToPlane(-1.0)

        def compute_rmse(self, s, t, c):
            return self.inner.compute_rmse(s, t, c)

        def compute_transformation(self, s, t, c):
            return self.inner.compute_transformation(s, t, c)

    res2 = cph.registration.registration_icp(source, target, d["max_dist"], np.eye(4, dtype=np.float32), Wrapped(),
                                             cph.registration.ICPConvergenceCriteria(max_iteration=12))
    o = orc.registration_icp(d["src"], d["tgt"], d["max_dist"], est=orc.EST_PT2PL, tgt_nrm=d["tgt_nrm"],
                             det_thresh=-1.0, max_iteration=12)
    # the generic loop transforms the estimator's copy incrementally (as the reference does), the
    # fused one applies the composed T: same answer to the path's tolerance
    assert np.linalg.norm(res2.transformation - o.transformation) <= 1e-5


def test_subclass_of_a_builtin_estimator_that_overrides_compute_transformation_is_called():
    """the fused device loop may only replace the reference loop for the built-in
    ComputeTransformation; an override must be called every iteration (registration.cu:157)."""
    cph, d, source, target = clouds(20000, seed=5)
    calls = []

    class Counting(cph.registration.TransformationEstimationPointToPlane):
        def compute_transformation(self, s, t, c):
            calls.append(len(np.asarray(c)))
            return super().compute_transformation(s, t, c)

    res = cph.registration.registration_icp(source, target, d["max_dist"], np.eye(4, dtype=np.float32),
                                            Counting(-1.0), cph.registration.ICPConvergenceCriteria(max_iteration=6))
    assert len(calls) >= 2 and np.linalg.norm(res.transformation - d["T_gt"]) < 1e-4
    # and a user estimator that calls estimate_normals (which used to wipe the engine's target)
    class Renormalising(cph.registration.TransformationEstimationPointToPlane):
        def compute_transformation(self, s, t, c):
            tmp = cph.geometry.PointCloud(np.asarray(s.points.cpu())[:2000])
            tmp.estimate_normals(cph.geometry.KDTreeSearchParamKNN(10))
            return super().compute_transformation(s, t, c)

    res2 = cph.registration.registration_icp(source, target, d["max_dist"], np.eye(4, dtype=np.float32),
                                             Renormalising(-1.0), cph.registration.ICPConvergenceCriteria(max_iteration=6))
    assert res2.fitness > 0.999 and np.linalg.norm(res2.transformation - res.transformation) < 1e-6


def test_pointcloud_members_and_dlpack_bridge(tmp_path):
    cph, d, source, target = clouds(20000, seed=2)
    T = d["T_gt"]
    moved = target.clone().transform(T)
    np.testing.assert_allclose(np.asarray(moved.points.cpu()), orc.transform_points(T, d["tgt"]), atol=1e-6)
    assert moved.has_normals() and not moved.has_colors() and not cph.geometry.PointCloud().has_points()
    down = target.voxel_down_sample(0.05)
    op, on, _ = orc.voxel_downsample(d["tgt"], 0.05, normals=d["tgt_nrm"])
    np.testing.assert_allclose(np.asarray(down.points.cpu()), op, atol=1e-6)
    np.testing.assert_allclose(np.asarray(down.normals.cpu()), on, atol=1e-5)
    assert len(target.voxel_down_sample(0.0).points) == 0          # warning + empty cloud (down_sample.cu:173-176)
    t = torch.from_dlpack(target.points)                           # zero copy out ...
    assert t.is_cuda and t.shape == (20000, 3) and t.data_ptr() == target.points.tensor.data_ptr()
    v = cph.utility.Vector3fVector.from_dlpack(t * 2.0)            # ... and in
    np.testing.assert_array_equal(np.asarray(v.cpu()), d["tgt"] * 2.0)
    assert len(v) == 20000 and v.size() == 20000
    # files on either side of the path
    cph.io.write_point_cloud(str(tmp_path / "t.ply"), target)
    cph.io.write_point_cloud(str(tmp_path / "t.pcd"), target)
    for name in ("t.ply", "t.pcd"):
        back = cph.io.read_point_cloud(str(tmp_path / name))
        np.testing.assert_array_equal(np.asarray(back.points.cpu()), d["tgt"])
        np.testing.assert_array_equal(np.asarray(back.normals.cpu()), d["tgt_nrm"])


def test_geometry_base_3d_members_of_the_cloud():
    """get_min_bound / get_max_bound / get_center / get_axis_aligned_bounding_box / translate / scale /
    rotate (geometry_base.h:44-90; geometry_utils.cu:150-270) against numpy in the functors' order."""
    cph, d, source, target = clouds(30000, seed=4)
    P, Nn = d["tgt"], d["tgt_nrm"]
    np.testing.assert_array_equal(target.get_min_bound(), P.min(0))
    np.testing.assert_array_equal(target.get_max_bound(), P.max(0))
    c = target.get_center()
    np.testing.assert_allclose(c, P.astype(np.float64).mean(0), atol=1e-7)
    box = target.get_axis_aligned_bounding_box()
    np.testing.assert_array_equal(box.get_extent(), P.max(0) - P.min(0))
    assert box.volume() == pytest.approx(float(np.prod(P.max(0) - P.min(0))))
    t = np.array([0.5, -1.0, 2.0], np.float32)
    moved = target.clone().translate(t)
    np.testing.assert_array_equal(np.asarray(moved.points.cpu()), P + t)                      # pt += t, exactly
    moved = target.clone().translate(t, relative=False)
    np.testing.assert_allclose(moved.get_center(), t, atol=3e-7)
    sc = target.clone().scale(3.0)
    np.testing.assert_array_equal(np.asarray(sc.points.cpu()), (P - c) * np.float32(3.0) + c)   # (pt - c) * s + c
    sc0 = target.clone().scale(3.0, center=False)
    np.testing.assert_array_equal(np.asarray(sc0.points.cpu()), P * np.float32(3.0))
    a = 0.4
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    rot = target.clone().rotate(R)
    want = (P - c).astype(np.float64) @ R.astype(np.float64).T + c
    np.testing.assert_allclose(np.asarray(rot.points.cpu()), want, atol=2e-7)
    np.testing.assert_allclose(np.asarray(rot.normals.cpu()), Nn.astype(np.float64) @ R.astype(np.float64).T, atol=2e-7)
    np.testing.assert_allclose(rot.get_center(), c, atol=3e-7)                                 # rotation about the centre
    empty = cph.geometry.PointCloud()
    assert not empty.get_center().any() and not empty.get_min_bound().any()


def test_error_conventions_of_the_reference():
    cph, d, source, target = clouds(5000, seed=6)
    bare = cph.geometry.PointCloud(d["tgt"])
    init = np.eye(4, dtype=np.float32)
    init[:3, 3] = [0.01, 0, 0]
    # missing normals: logged error, identity updates -> init comes back (registration.cu:134-143)
    res = cph.registration.registration_icp(source, bare, d["max_dist"], init,
                                            cph.registration.TransformationEstimationPointToPlane())
    np.testing.assert_array_equal(res.transformation, init)
    # invalid distance: logged error, empty result
    res = cph.registration.registration_icp(source, target, 0.0, init)
    assert res.fitness == 0.0 and len(res.correspondence_set) == 0
    np.testing.assert_array_equal(res.transformation, init)
    # empty clouds
    res = cph.registration.evaluate_registration(cph.geometry.PointCloud(), target, 0.1)
    assert res.fitness == 0.0 and len(res.correspondence_set) == 0


def test_iteration_callback_reports_what_the_reference_logs():
    """mi_icp_set_iteration_callback: once per iteration, in order, the fitness / inlier RMSE of the evaluation
    the iteration's update starts from (registration.cu:155-156) -- the values the stepping interface returns
    evaluation by evaluation; nothing is reported once the callback is removed."""
    from conftest import make_pair
    from cupoch_amd.engine import Engine
    d = make_pair(30000, seed=9, noise=0.05)
    eng = Engine(0)
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.set_source(d["src"])
    seen = []
    eng.set_iteration_callback(lambda i, f, r: seen.append((i, f, r)))
    res = eng.registration_icp(2, d["max_dist"], None, 0.0, 0.0, 21, -1.0)
    assert res.iterations == 21 and [s[0] for s in seen] == list(range(21))
    eng.set_iteration_callback(None)
    step = [eng.icp_begin(2, d["max_dist"], None, -1.0)]
    for _ in range(20):
        step.append(eng.icp_iterate(1))
    assert len(seen) == 21                                   # (no callback any more)
    for (i, f, r), s in zip(seen, step):
        assert f == s.fitness and r == s.inlier_rmse, i
    # a loop that converges early reports the iterations it ran
    seen.clear()
    eng.set_iteration_callback(lambda i, f, r: seen.append(i))
    res = eng.registration_icp(2, d["max_dist"], None, 1e-6, 1e-6, 30, -1.0)
    assert seen == list(range(res.iterations)) and res.iterations < 30
    eng.close()
