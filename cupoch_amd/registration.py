"""cupoch.registration mirror: same names, defaults and semantics as
src/cupoch/registration/{registration,transformation_estimation,generalized_icp}.h
and src/python/cupoch_pybind/registration/registration.cpp:62-478, executed by
the HIP engine through the C ABI (include/mi_icp.h)."""
import copy
import enum

import numpy as np

from . import _lib
from .engine import kabsch_from_sums, solve_system
from .geometry import PointCloud, get_engine


class TransformationEstimationType(enum.IntEnum):
    # registration/transformation_estimation.h:38-45
    Unspecified = 0
    PointToPoint = 1
    PointToPlane = 2
    SymmetricMethod = 3
    ColoredICP = 4
    GeneralizedICP = 5


class ICPConvergenceCriteria:
    # registration/registration.h:35-49
    def __init__(self, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
        self.relative_fitness = float(relative_fitness)
        self.relative_rmse = float(relative_rmse)
        self.max_iteration = int(max_iteration)

    def __repr__(self):
        return ("registration::ICPConvergenceCriteria class with relative_fitness={:e}, "
                "relative_rmse={:e}, and max_iteration={:d}").format(
                    self.relative_fitness, self.relative_rmse, self.max_iteration)


class RegistrationResult:
    # registration/registration.h:51-67
    def __init__(self, transformation=None):
        self.transformation = (np.eye(4, dtype=np.float32) if transformation is None
                               else np.asarray(transformation, np.float32).copy())
        self.correspondence_set = np.zeros((0, 2), np.int32)
        self.inlier_rmse = 0.0
        self.fitness = 0.0

    def __repr__(self):
        return ("registration::RegistrationResult with fitness={:f}, inlier_rmse={:f}, and "
                "correspondence_set size of {:d}").format(self.fitness, self.inlier_rmse,
                                                          len(self.correspondence_set))


class TransformationEstimation:
    """Abstract base (transformation_estimation.h:50-65).  Subclass it in Python and
    registration_icp falls back to the generic loop that calls your
    compute_transformation every iteration (the pybind trampoline,
    registration.cpp:34-60)."""
    _est = None

    def get_transformation_estimation_type(self):
        return TransformationEstimationType.Unspecified

    def compute_rmse(self, source, target, corres):
        raise NotImplementedError

    def compute_transformation(self, source, target, corres):
        raise NotImplementedError


class _BuiltinEstimation(TransformationEstimation):
    det_thresh = -1.0

    def get_transformation_estimation_type(self):
        return TransformationEstimationType(self._est)

    def _prepare(self, source, target, corres):
        eng = get_engine()
        _load_clouds(eng, source, target)
        eng.set_correspondences(np.asarray(_corres_host(corres), np.int32).reshape(-1, 2))
        return eng

    def compute_rmse(self, source, target, corres):
        if len(_corres_host(corres)) == 0:
            return 0.0
        return self._prepare(source, target, corres).compute_rmse(self._est)

    def compute_transformation(self, source, target, corres):
        if len(_corres_host(corres)) == 0:
            return np.eye(4, dtype=np.float32)
        return self._prepare(source, target, corres).compute_transformation(
            self._est, det_thresh=self.det_thresh)


class TransformationEstimationPointToPoint(_BuiltinEstimation):
    _est = _lib.EST_POINT_TO_POINT


class TransformationEstimationPointToPlane(_BuiltinEstimation):
    _est = _lib.EST_POINT_TO_PLANE

    def __init__(self, det_thresh=1.0e-6):   # transformation_estimation.h:94
        self.det_thresh = float(det_thresh)


class TransformationEstimationSymmetricMethod(_BuiltinEstimation):
    _est = _lib.EST_SYMMETRIC

    def __init__(self, det_thresh=1.0e-6):   # transformation_estimation.h:121
        self.det_thresh = float(det_thresh)


class TransformationEstimationForGeneralizedICP(_BuiltinEstimation):
    _est = _lib.EST_GENERALIZED

    def __init__(self, epsilon=1e-3):        # generalized_icp.h:20
        self.epsilon = float(epsilon)


def _corres_host(corres):
    if hasattr(corres, "cpu"):
        return np.asarray(corres.cpu())
    return np.asarray(corres)


def _t(v):
    return v.tensor if v is not None and len(v) else None


def _load_clouds(eng, source, target):
    eng.set_target(target.points.tensor, _t(target._normals) if target.has_normals() else None,
                   _t(target._covariances) if target.has_covariances() else None)
    eng.set_source(source.points.tensor, _t(source._normals) if source.has_normals() else None,
                   _t(source._covariances) if source.has_covariances() else None)


def _result_from(eng, res):
    out = RegistrationResult(np.array(res.transformation, np.float32).reshape(4, 4).T)
    out.fitness = float(res.fitness)
    out.inlier_rmse = float(res.inlier_rmse)
    out.correspondence_set = eng.get_correspondences()
    return out


def evaluate_registration(source, target, max_correspondence_distance, transformation=None):
    """registration::EvaluateRegistration (registration.cu:106-119)"""
    eng = get_engine()
    _load_clouds(eng, source, target)
    T = np.eye(4, dtype=np.float32) if transformation is None else transformation
    return _result_from(eng, eng.evaluate_registration(max_correspondence_distance, T))


def registration_icp(source, target, max_correspondence_distance, init=None,
                     estimation_method=None, criteria=None):
    """registration::RegistrationICP (registration.cu:121-172)"""
    init = np.eye(4, dtype=np.float32) if init is None else np.asarray(init, np.float32)
    est = TransformationEstimationPointToPoint() if estimation_method is None else estimation_method
    crit = ICPConvergenceCriteria() if criteria is None else criteria
    if max_correspondence_distance <= 0.0:
        print("[cupoch_amd] Error: Invalid max_correspondence_distance.")   # LogError, keeps going
    if (est.get_transformation_estimation_type() in (TransformationEstimationType.PointToPlane,
                                                      TransformationEstimationType.ColoredICP)
            and not target.has_normals()):
        print("[cupoch_amd] Error: TransformationEstimationPointToPlane and "
              "TransformationEstimationColoredICP require pre-computed target normal vectors.")
    if _is_builtin(est):
        eng = get_engine()
        _load_clouds(eng, source, target)
        res = eng.registration_icp(est._est, max_correspondence_distance, init,
                                   crit.relative_fitness, crit.relative_rmse, crit.max_iteration,
                                   getattr(est, "det_thresh", -1.0))
        return _result_from(eng, res)
    return _generic_icp(source, target, max_correspondence_distance, init, est, crit)


def _is_builtin(est):
    """The device-resident loop may replace the reference's loop only while the estimator's
    ComputeTransformation is the built-in one: a user subclass that overrides it is called
    every iteration, as the reference's virtual call would (registration.cu:157)."""
    return (isinstance(est, _BuiltinEstimation)
            and type(est).compute_transformation is _BuiltinEstimation.compute_transformation)


def _generic_icp(source, target, max_dist, init, est, crit):
    """The reference loop, for user-defined estimators.  The engine keeps the ORIGINAL source
    and the target tree and evaluates under the accumulated transformation (seeded by the
    previous iteration's matches); the estimator still sees the transformed copy.  If the
    estimator itself goes through the engine, the clouds are simply loaded again."""
    eng = get_engine()
    pcd = source.clone()
    transformation = init.copy()
    if not np.allclose(init, np.eye(4), atol=1e-5, rtol=0):
        pcd.transform(init)
    loaded = [None]

    def evaluate(T):
        if loaded[0] != getattr(eng, "generation", 0):
            _load_clouds(eng, source, target)
            loaded[0] = eng.generation
        out = _result_from(eng, eng.evaluate_registration(max_dist, T))
        out.transformation = T.copy()
        return out

    result = evaluate(transformation)
    for _ in range(crit.max_iteration):
        update = np.asarray(est.compute_transformation(pcd, target, result.correspondence_set),
                            np.float32)
        transformation = (update @ transformation).astype(np.float32)
        pcd.transform(update)
        backup = result
        result = evaluate(transformation)
        if (abs(backup.fitness - result.fitness) < crit.relative_fitness and
                abs(backup.inlier_rmse - result.inlier_rmse) < crit.relative_rmse):
            break
    return result


def registration_colored_icp(source, target, max_correspondence_distance, init=None,
                             criteria=None, lambda_geometric=0.968, det_thresh=1.0e-6):
    """registration::RegistrationColoredICP (colored_icp.cu:329-341; pybind signature
    registration.cpp:433-439).  The estimator class itself is private to the reference's
    translation unit, so only this function is mirrored."""
    init = np.eye(4, dtype=np.float32) if init is None else np.asarray(init, np.float32)
    crit = ICPConvergenceCriteria() if criteria is None else criteria
    if max_correspondence_distance <= 0.0:
        print("[cupoch_amd] Error: Invalid max_correspondence_distance.")
    if not target.has_normals():
        print("[cupoch_amd] Error: TransformationEstimationPointToPlane and "
              "TransformationEstimationColoredICP require pre-computed target normal vectors.")
    eng = get_engine()
    _load_clouds(eng, source, target)
    if target.has_normals() and target.has_colors():
        eng.set_target_colors(target.colors.tensor)
    if source.has_colors():
        eng.set_source_colors(source.colors.tensor)
    res = eng.registration_colored_icp(max_correspondence_distance, init, crit.relative_fitness,
                                       crit.relative_rmse, crit.max_iteration, lambda_geometric,
                                       det_thresh)
    return _result_from(eng, res)


def _initialize_for_gicp(pcd, epsilon):
    """InitializePointCloudForGeneralizedICP (generalized_icp.cu:37-61)"""
    out = pcd.clone()
    if out.has_covariances():
        return out
    if not out.has_normals():
        from .geometry import KDTreeSearchParamKNN
        out.estimate_normals(KDTreeSearchParamKNN(20))
    eng = get_engine()
    out.covariances = eng.covariances_from_normals(out.normals.tensor, epsilon)
    return out


def registration_generalized_icp(source, target, max_correspondence_distance, init=None,
                                 estimation=None, criteria=None):
    """registration::RegistrationGeneralizedICP (generalized_icp.cu:185-198)"""
    est = TransformationEstimationForGeneralizedICP() if estimation is None else estimation
    return registration_icp(_initialize_for_gicp(source, est.epsilon),
                            _initialize_for_gicp(target, est.epsilon),
                            max_correspondence_distance, init, est, criteria)
