"""cupoch.odometry mirror (src/cupoch/odometry/odometry.h:43-61, odometry_option.h:30-62,
rgbdodometry_jacobian.h:33-134; python surface src/python/cupoch_pybind/odometry/odometry.cpp):
compute_rgbd_odometry and compute_weighted_rgbd_odometry over two RGB-D frames."""
import numpy as np

from . import geometry


class OdometryOption:
    def __init__(self, iteration_number_per_pyramid_level=(20, 10, 5), max_depth_diff=0.03, min_depth=0.0,
                 max_depth=4.0, nu=5.0, sigma2_init=1.0, inv_sigma_mat_diag=None):
        self.iteration_number_per_pyramid_level = list(iteration_number_per_pyramid_level)
        self.max_depth_diff = float(max_depth_diff)
        self.min_depth = float(min_depth)
        self.max_depth = float(max_depth)
        self.nu = float(nu)
        self.sigma2_init = float(sigma2_init)
        self.inv_sigma_mat_diag = np.zeros(6, np.float32) if inv_sigma_mat_diag is None else np.asarray(inv_sigma_mat_diag, np.float32)


class RGBDOdometryJacobian:
    COLOR_TERM, HYBRID_TERM = 0, 1
    jacobian_type = None


class RGBDOdometryJacobianFromColorTerm(RGBDOdometryJacobian):
    jacobian_type = RGBDOdometryJacobian.COLOR_TERM


class RGBDOdometryJacobianFromHybridTerm(RGBDOdometryJacobian):
    jacobian_type = RGBDOdometryJacobian.HYBRID_TERM


def compute_rgbd_odometry(source, target, pinhole_camera_intrinsic, odo_init=None,
                          jacobian=None, option=None):
    """odometry::ComputeRGBDOdometry.  source / target: geometry.RGBDImage with a float32 intensity
    image (.color) and a float32 depth image (.depth).  Returns (success, 4x4 transformation mapping
    the source frame onto the target frame, 6x6 information matrix)."""
    jacobian = RGBDOdometryJacobianFromHybridTerm() if jacobian is None else jacobian
    option = OdometryOption() if option is None else option
    init = np.eye(4, dtype=np.float32) if odo_init is None else np.asarray(odo_init, np.float32)
    imgs = (source.color, source.depth, target.color, target.depth)
    shapes = {tuple(np.shape(x)) for x in imgs}
    ok_types = all(str(getattr(x, "dtype", "")).replace("torch.", "") == "float32" for x in imgs)
    if len(shapes) != 1 or len(next(iter(shapes))) != 2 or not ok_types:      # CheckRGBDImagePair (odometry.cu:482-496)
        print("[cupoch_amd] Warning: [RGBDOdometry] Two RGBD pairs should be same in size.")
        return False, np.eye(4, dtype=np.float32), np.zeros((6, 6), np.float64)
    d = imgs[0]
    dev = d.device.index if (hasattr(d, "is_cuda") and d.is_cuda) else None
    return geometry.get_engine(dev).compute_rgbd_odometry(
        imgs[0], imgs[1], imgs[2], imgs[3], pinhole_camera_intrinsic.as4(), init, jacobian.jacobian_type,
        option.iteration_number_per_pyramid_level, option.max_depth_diff, option.min_depth, option.max_depth)


def compute_weighted_rgbd_odometry(source, target, pinhole_camera_intrinsic, odo_init=None, prev_twist=None,
                                   jacobian=None, option=None):
    """odometry::ComputeWeightedRGBDOdometry (always the hybrid term, odometry.cu:937-941).
    Returns (success, transformation, twist (6), information)."""
    option = OdometryOption() if option is None else option
    init = np.eye(4, dtype=np.float32) if odo_init is None else np.asarray(odo_init, np.float32)
    imgs = (source.color, source.depth, target.color, target.depth)
    shapes = {tuple(np.shape(x)) for x in imgs}
    ok_types = all(str(getattr(x, "dtype", "")).replace("torch.", "") == "float32" for x in imgs)
    if len(shapes) != 1 or len(next(iter(shapes))) != 2 or not ok_types:
        print("[cupoch_amd] Warning: [RGBDOdometry] Two RGBD pairs should be same in size.")
        return False, np.eye(4, dtype=np.float32), np.zeros(6, np.float32), np.zeros((6, 6), np.float64)
    d = imgs[0]
    dev = d.device.index if (hasattr(d, "is_cuda") and d.is_cuda) else None
    return geometry.get_engine(dev).compute_rgbd_odometry(
        imgs[0], imgs[1], imgs[2], imgs[3], pinhole_camera_intrinsic.as4(), init, 1,
        option.iteration_number_per_pyramid_level, option.max_depth_diff, option.min_depth, option.max_depth,
        True, prev_twist, option.nu, option.sigma2_init, option.inv_sigma_mat_diag)
