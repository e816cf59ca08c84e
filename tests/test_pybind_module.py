"""The reference's pybind11 module built on the C++ surface (cupoch_amd/cpp/src/pybind_module.cpp):
names and defaults of cupoch_pybind (CPU: import + surface; GPU: same results as the ctypes mirror)."""
import numpy as np
import pytest

from conftest import make_pair


def test_module_surface_matches_the_reference_names_and_defaults():
    from cupoch_amd import pybind as cph
    cph.initialize_allocator()                                   # cupoch/__init__.py calls this on import
    r = cph.registration
    c = r.ICPConvergenceCriteria()                               # registration.h:37-39
    assert (c.relative_fitness, c.relative_rmse, c.max_iteration) == (pytest.approx(1e-6), pytest.approx(1e-6), 30)
    assert r.TransformationEstimationPointToPlane().det_thresh == pytest.approx(1e-6)
    assert r.TransformationEstimationSymmetricMethod().det_thresh == pytest.approx(1e-6)
    assert r.TransformationEstimationForGeneralizedICP().epsilon == pytest.approx(1e-3)
    T = r.TransformationEstimationType                           # transformation_estimation.h:38-45
    assert [int(T.Unspecified), int(T.PointToPoint), int(T.PointToPlane), int(T.SymmetricMethod), int(T.ColoredICP),
            int(T.GeneralizedICP)] == [0, 1, 2, 3, 4, 5]
    assert r.TransformationEstimationPointToPlane().get_transformation_estimation_type() == T.PointToPlane
    for name in ("registration_icp", "evaluate_registration", "registration_generalized_icp",
                 "registration_colored_icp", "RegistrationResult"):
        assert hasattr(r, name)
    g = cph.geometry
    assert g.KDTreeSearchParamKNN().knn == 30 and g.KDTreeSearchParamRadius(0.1, 20).max_nn == 20
    res = r.RegistrationResult()
    assert np.array_equal(res.transformation, np.eye(4, dtype=np.float32)) and res.correspondence_set.shape == (0, 2)
    with pytest.raises(TypeError):
        g.KDTreeSearchParamRadius(0.1)                           # no default max_nn in the reference either


@pytest.mark.gpu
def test_registration_through_the_pybind_module_equals_the_ctypes_mirror():
    from cupoch_amd import geometry, pybind as cph, registration, utility
    d = make_pair(60000, seed=21, noise=0.02)
    src, tgt = cph.geometry.PointCloud(), cph.geometry.PointCloud()
    src.points = cph.utility.Vector3fVector(d["src"])            # the reference's idiom (numpy_interop.py)
    tgt.points = cph.utility.Vector3fVector(d["tgt"])
    tgt.normals = d["tgt_nrm"]                                    # array-likes are accepted as well
    assert np.array_equal(np.asarray(src.points.cpu()), d["src"]) and tgt.has_normals() and not src.has_normals()
    init = np.eye(4, dtype=np.float32)
    crit = cph.registration.ICPConvergenceCriteria(max_iteration=12)
    got = {}
    for name, est in (("p2p", cph.registration.TransformationEstimationPointToPoint()),
                      ("pt2pl", cph.registration.TransformationEstimationPointToPlane())):
        got[name] = cph.registration.registration_icp(src, tgt, d["max_dist"], init, est, crit)
    ev = cph.registration.evaluate_registration(src, tgt, d["max_dist"])
    # the ctypes mirror on the same clouds
    s2, t2 = geometry.PointCloud(), geometry.PointCloud()
    s2.points, t2.points = utility.Vector3fVector(d["src"]), utility.Vector3fVector(d["tgt"])
    t2.normals = utility.Vector3fVector(d["tgt_nrm"])
    c2 = registration.ICPConvergenceCriteria(max_iteration=12)
    for name, est in (("p2p", registration.TransformationEstimationPointToPoint()),
                      ("pt2pl", registration.TransformationEstimationPointToPlane())):
        ref = registration.registration_icp(s2, t2, d["max_dist"], init, est, c2)
        r = got[name]
        assert np.linalg.norm(r.transformation - np.asarray(ref.transformation, np.float32)) <= 1e-6, name
        assert r.fitness == pytest.approx(ref.fitness, abs=1e-6) and r.inlier_rmse == pytest.approx(ref.inlier_rmse, rel=1e-5)
        assert np.array_equal(r.correspondence_set, np.asarray(ref.correspondence_set, np.int32).reshape(-1, 2)), name
        assert np.linalg.norm(r.transformation - d["T_gt"]) < 5e-3
    e2 = registration.evaluate_registration(s2, t2, d["max_dist"])
    assert ev.fitness == pytest.approx(e2.fitness, abs=1e-6) and ev.inlier_rmse == pytest.approx(e2.inlier_rmse, rel=1e-5)
    # geometry members through the module
    moved = cph.geometry.PointCloud(d["src"]).transform(d["T_gt"].astype(np.float32))
    assert np.abs(np.asarray(moved.points.cpu()).mean(0) - d["tgt"].mean(0)).max() < 0.05
    down = tgt.voxel_down_sample(0.05)
    assert 0 < len(down) < len(tgt) and down.has_normals()
    assert np.all(np.asarray(tgt.get_min_bound()) <= np.asarray(tgt.get_center()))
