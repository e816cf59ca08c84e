"""ctypes front-end of the CPU oracle (oracle/icp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by anything under cupoch_amd/.

All matrices cross this API as numpy row-major (what a user writes); the C
side uses Eigen's column-major layout, so 4x4 transforms are transposed here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

EST_P2P, EST_PT2PL, EST_SYM, EST_GICP = 1, 2, 3, 5
EST_COLORED = 4


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    srcs = [os.path.join(_HERE, f) for f in ("icp_oracle.c", "odometry_oracle.c")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    ref = os.environ.get("CUPOCH_REFERENCE", "/root/reference")
    if os.path.isdir(os.path.join(ref, "src", "tests", "test_utility")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_ref", "REF=" + ref])


class _Result(C.Structure):
    _fields_ = [("transformation", C.c_float * 16),
                ("fitness", C.c_float),
                ("inlier_rmse", C.c_float),
                ("n_corres", C.c_int64),
                ("iterations", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.oracle_search_knn.restype = C.c_int64
        _lib.oracle_search_radius.restype = C.c_int64
        _lib.oracle_search_bruteforce.restype = C.c_int64
        _lib.oracle_voxel_downsample.restype = C.c_int64
        _lib.oracle_create_from_depth.restype = C.c_int64
        _lib.oracle_od_correspondence.restype = C.c_int64
        _lib.oracle_compute_rmse.restype = C.c_float
        _lib.oracle_tree_create.restype = C.c_void_p
        _lib.oracle_tree_search_radius.restype = C.c_int64
        _lib.oracle_tree_search_knn.restype = C.c_int64
    return _lib


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _T_in(T):
    """row-major numpy 4x4 -> column-major float[16]"""
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(4, 4).T)


def _T_out(buf):
    return np.array(buf, dtype=np.float32).reshape(4, 4).T.copy()


def num_threads():
    return int(lib().oracle_num_threads())


def transform_points(T, pts):
    out = _f32(pts, (-1, 3)).copy()
    lib().oracle_transform_points(_p(_T_in(T)), _p(out), C.c_int64(len(out)))
    return out


def transform_normals(T, nrm):
    out = _f32(nrm, (-1, 3)).copy()
    lib().oracle_transform_normals(_p(_T_in(T)), _p(out), C.c_int64(len(out)))
    return out


def rotate_covariances(T, covs):
    """covs: (n,3,3) row-major symmetric or general; returns R C R^T."""
    c = _f32(covs, (-1, 3, 3))
    cm = np.ascontiguousarray(c.transpose(0, 2, 1))  # column-major per matrix
    lib().oracle_rotate_covariances(_p(_T_in(T)), _p(cm), C.c_int64(len(cm)))
    return np.ascontiguousarray(cm.transpose(0, 2, 1))


def search_knn(tgt, qry, k):
    tgt, qry = _f32(tgt, (-1, 3)), _f32(qry, (-1, 3))
    idx = np.empty((len(qry), max(k, 1)), np.int32)
    d2 = np.empty((len(qry), max(k, 1)), np.float32)
    r = lib().oracle_search_knn(_p(tgt), C.c_int64(len(tgt)), _p(qry), C.c_int64(len(qry)),
                                C.c_int(k), _p(idx), _p(d2))
    return int(r), idx, d2


def search_radius(tgt, qry, radius, max_nn):
    tgt, qry = _f32(tgt, (-1, 3)), _f32(qry, (-1, 3))
    idx = np.empty((len(qry), max(max_nn, 1)), np.int32)
    d2 = np.empty((len(qry), max(max_nn, 1)), np.float32)
    r = lib().oracle_search_radius(_p(tgt), C.c_int64(len(tgt)), _p(qry), C.c_int64(len(qry)),
                                   C.c_float(radius), C.c_int(max_nn), _p(idx), _p(d2))
    return int(r), idx, d2


class Tree:
    """The oracle's kd-tree over one target, kept across searches (large clouds)."""

    def __init__(self, tgt):
        self.tgt = _f32(tgt, (-1, 3))
        self.h = C.c_void_p(lib().oracle_tree_create(_p(self.tgt), C.c_int64(len(self.tgt))))

    def search_radius(self, qry, radius, max_nn=1):
        qry = _f32(qry, (-1, 3))
        idx = np.empty((len(qry), max(max_nn, 1)), np.int32)
        d2 = np.empty((len(qry), max(max_nn, 1)), np.float32)
        r = lib().oracle_tree_search_radius(self.h, _p(qry), C.c_int64(len(qry)), C.c_float(radius),
                                            C.c_int(max_nn), _p(idx), _p(d2))
        return int(r), idx, d2

    def search_knn(self, qry, k):
        qry = _f32(qry, (-1, 3))
        idx = np.empty((len(qry), max(k, 1)), np.int32)
        d2 = np.empty((len(qry), max(k, 1)), np.float32)
        r = lib().oracle_tree_search_knn(self.h, _p(qry), C.c_int64(len(qry)), C.c_int(k), _p(idx), _p(d2))
        return int(r), idx, d2

    def close(self):
        if self.h:
            lib().oracle_tree_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def search_bruteforce(tgt, qry, k, radius=-1.0):
    tgt, qry = _f32(tgt, (-1, 3)), _f32(qry, (-1, 3))
    idx = np.empty((len(qry), k), np.int32)
    d2 = np.empty((len(qry), k), np.float32)
    r = lib().oracle_search_bruteforce(_p(tgt), C.c_int64(len(tgt)), _p(qry),
                                       C.c_int64(len(qry)), C.c_int(k), C.c_float(radius),
                                       _p(idx), _p(d2))
    return int(r), idx, d2


def _cov_cm(covs):
    if covs is None:
        return None
    c = _f32(covs, (-1, 3, 3))
    return np.ascontiguousarray(c.transpose(0, 2, 1))


def compute_system(est, src, tgt, corres, src_nrm=None, tgt_nrm=None, src_cov=None,
                   tgt_cov=None):
    """The accumulated 32-double system (layout documented in icp_oracle.c)."""
    src, tgt = _f32(src, (-1, 3)), _f32(tgt, (-1, 3))
    sn, tn = _f32(src_nrm, (-1, 3)), _f32(tgt_nrm, (-1, 3))
    sc, tc = _cov_cm(src_cov), _cov_cm(tgt_cov)
    cor = np.ascontiguousarray(corres, dtype=np.int32).reshape(-1, 2)
    sys = np.zeros(32, np.float64)
    lib().oracle_compute_system(C.c_int(est), _p(src), _p(sn), _p(sc), _p(tgt), _p(tn), _p(tc),
                                _p(cor), C.c_int64(len(cor)), _p(sys))
    return sys


def solve_system(sys, det_thresh):
    T = np.zeros(16, np.float32)
    sys = np.ascontiguousarray(sys, np.float64)
    ok = lib().oracle_solve_system(_p(sys), C.c_float(det_thresh), _p(T))
    return bool(ok), T.reshape(4, 4).T.copy()


def kabsch_from_sums(sys, n_model):
    T = np.zeros(16, np.float32)
    sys = np.ascontiguousarray(sys, np.float64)
    lib().oracle_kabsch_from_sums(_p(sys), C.c_int64(n_model), _p(T))
    return T.reshape(4, 4).T.copy()


def kabsch(model, target):
    model, target = _f32(model, (-1, 3)), _f32(target, (-1, 3))
    T = np.zeros(16, np.float32)
    lib().oracle_kabsch(_p(model), _p(target), C.c_int64(len(model)), _p(T))
    return T.reshape(4, 4).T.copy()


def vector6_to_matrix4(x):
    x = _f32(x, (6,))
    T = np.zeros(16, np.float32)
    lib().oracle_vector6_to_matrix4(_p(x), _p(T))
    return T.reshape(4, 4).T.copy()


def compute_rmse(est, src, tgt, corres, src_nrm=None, tgt_nrm=None, src_cov=None, tgt_cov=None):
    src, tgt = _f32(src, (-1, 3)), _f32(tgt, (-1, 3))
    sn, tn = _f32(src_nrm, (-1, 3)), _f32(tgt_nrm, (-1, 3))
    sc, tc = _cov_cm(src_cov), _cov_cm(tgt_cov)
    cor = np.ascontiguousarray(corres, dtype=np.int32).reshape(-1, 2)
    return float(lib().oracle_compute_rmse(C.c_int(est), _p(src), _p(sn), _p(sc), _p(tgt),
                                           _p(tn), _p(tc), _p(cor), C.c_int64(len(cor))))


def covariances_from_normals(nrm, eps=1e-3):
    nrm = _f32(nrm, (-1, 3))
    cm = np.empty((len(nrm), 3, 3), np.float32)
    lib().oracle_covariances_from_normals(_p(nrm), C.c_int64(len(nrm)), C.c_float(eps), _p(cm))
    return np.ascontiguousarray(cm.transpose(0, 2, 1))


def gicp_weight(Cs, Ct):
    W = np.zeros(9, np.float32)
    cs = np.ascontiguousarray(np.asarray(Cs, np.float32).reshape(3, 3).T)
    ct = np.ascontiguousarray(np.asarray(Ct, np.float32).reshape(3, 3).T)
    lib().oracle_gicp_weight(_p(cs), _p(ct), _p(W))
    return W.reshape(3, 3)


class RegistrationResult:
    def __init__(self, res, corres):
        self.transformation = _T_out(res.transformation)
        self.fitness = float(res.fitness)
        self.inlier_rmse = float(res.inlier_rmse)
        self.correspondence_set = corres[: int(res.n_corres)].copy()
        self.iterations = int(res.iterations)


def evaluate_registration(src, tgt, max_dist, T=None):
    src, tgt = _f32(src, (-1, 3)), _f32(tgt, (-1, 3))
    T = np.eye(4, dtype=np.float32) if T is None else T
    cor = np.empty((max(len(src), 1), 2), np.int32)
    res = _Result()
    lib().oracle_evaluate_registration(_p(src), C.c_int64(len(src)), _p(tgt),
                                       C.c_int64(len(tgt)), C.c_float(max_dist),
                                       _p(_T_in(T)), _p(cor), C.byref(res))
    return RegistrationResult(res, cor)


def registration_icp(src, tgt, max_dist, init=None, est=EST_P2P, det_thresh=None,
                     relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30,
                     src_nrm=None, tgt_nrm=None, src_cov=None, tgt_cov=None, composed=False):
    """Restatement of registration::RegistrationICP (registration.cu:121-172).
    composed=True: DESIGN.md deviation 1 restated -- the composed transformation applied to the pristine
    source every iteration (what the engine does) instead of the reference's incremental fp32 transforms."""
    src, tgt = _f32(src, (-1, 3)), _f32(tgt, (-1, 3))
    sn, tn = _f32(src_nrm, (-1, 3)), _f32(tgt_nrm, (-1, 3))
    sc, tc = _cov_cm(src_cov), _cov_cm(tgt_cov)
    init = np.eye(4, dtype=np.float32) if init is None else init
    if det_thresh is None:
        det_thresh = 1e-6 if est in (EST_PT2PL, EST_SYM) else -1.0
    cor = np.empty((max(len(src), 1), 2), np.int32)
    res = _Result()
    lib().oracle_set_composed(C.c_int(1 if composed else 0))
    try:
        lib().oracle_registration_icp(
            _p(src), _p(sn), _p(sc), C.c_int64(len(src)), _p(tgt), _p(tn), _p(tc),
            C.c_int64(len(tgt)), C.c_float(max_dist), _p(_T_in(init)), C.c_int(est),
            C.c_float(det_thresh), C.c_float(relative_fitness), C.c_float(relative_rmse),
            C.c_int(max_iteration), _p(cor), C.byref(res))
    finally:
        lib().oracle_set_composed(C.c_int(0))
    return RegistrationResult(res, cor)


def voxel_downsample(pts, voxel, normals=None, colors=None):
    pts = _f32(pts, (-1, 3))
    nrm, col = _f32(normals, (-1, 3)), _f32(colors, (-1, 3))
    op = np.empty_like(pts)
    on = np.empty_like(pts) if nrm is not None else None
    oc = np.empty_like(pts) if col is not None else None
    m = lib().oracle_voxel_downsample(_p(pts), _p(nrm), _p(col), C.c_int64(len(pts)),
                                      C.c_float(voxel), _p(op), _p(on), _p(oc))
    m = int(m)
    return op[:m].copy(), (None if on is None else on[:m].copy()), \
        (None if oc is None else oc[:m].copy())


def estimate_normals_knn(pts, k=30):
    pts = _f32(pts, (-1, 3))
    out = np.empty_like(pts)
    lib().oracle_estimate_normals_knn(_p(pts), C.c_int64(len(pts)), C.c_int(k), _p(out))
    return out


_colored_keep = []


def intensity(colors):
    """(c0 + c1 + c2) / 3.0 as colored_icp.cu:91 evaluates it: fp32 sum, division in double."""
    c = np.asarray(colors, np.float32).reshape(-1, 3)
    s = (c[:, 0] + c[:, 1]) + c[:, 2]
    return _f32((s.astype(np.float64) / 3.0).astype(np.float32))


def set_colored_context(src_colors, tgt_colors, tgt_grad, lambda_geometric=0.968):
    """Inputs of the ColoredICP estimator for the following compute_system /
    compute_rmse / registration_icp(est=EST_COLORED) calls."""
    si, ti = intensity(src_colors), intensity(tgt_colors)
    tg = _f32(tgt_grad, (-1, 3))
    _colored_keep[:] = [si, ti, tg]
    lib().oracle_set_colored_context(_p(si), _p(ti), _p(tg), C.c_float(lambda_geometric))


def color_gradients(pts, nrm, colors, radius, max_nn=30):
    """InitializePointCloudForColoredICP (colored_icp.cu:108-148)"""
    pts, nrm = _f32(pts, (-1, 3)), _f32(nrm, (-1, 3))
    inten = intensity(colors)
    out = np.empty_like(pts)
    lib().oracle_color_gradients(_p(pts), _p(nrm), _p(inten), C.c_int64(len(pts)), C.c_float(radius),
                                 C.c_int(max_nn), _p(out))
    return out


def registration_colored_icp(src, tgt, max_dist, src_colors, tgt_colors, tgt_nrm, init=None,
                             lambda_geometric=0.968, det_thresh=1e-6, **kw):
    """registration::RegistrationColoredICP (colored_icp.cu:329-341)"""
    grad = color_gradients(tgt, tgt_nrm, tgt_colors, max_dist * 2.0, 30)
    set_colored_context(src_colors, tgt_colors, grad, lambda_geometric)
    return registration_icp(src, tgt, max_dist, init=init, est=EST_COLORED, det_thresh=det_thresh,
                            tgt_nrm=tgt_nrm, **kw)


def pyramid_level_intrinsic(width, height, fx, fy, cx, cy, level):
    """PinholeCameraIntrinsic::CreatePyramidLevel (camera/pinhole_camera_intrinsic.cpp:82-92)"""
    if level == 0 or width <= 0 or height <= 0:
        return width, height, np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy)
    s = np.float32(np.float32(0.5) ** np.float32(level))
    h = np.float32(0.5)
    return (width >> level, height >> level, np.float32(fx) * s, np.float32(fy) * s,
            (np.float32(cx) + h) * s - h, (np.float32(cy) + h) * s - h)


def create_from_depth(depth, intrinsic4, extrinsic=None, color=None, depth_scale=1000.0,
                      depth_trunc=1000.0, depth_cutoff=-1.0, stride=1, rgbd=False,
                      compute_normals=False, valid_only=True):
    """PointCloud::CreateFromDepthImage / CreateFromRGBDImage (pointcloud_factory.cu:286-376)."""
    depth = np.ascontiguousarray(depth)
    h, w = depth.shape
    if depth.dtype == np.uint16:
        f = np.empty((h, w), np.float32)
        lib().oracle_depth_u16_to_float(depth.ctypes.data_as(C.c_void_p), C.c_int64(h * w),
                                        C.c_float(depth_scale), C.c_float(depth_trunc), _p(f))
        depth = f
    depth = _f32(depth)
    kind = 0
    if color is not None:
        color = np.ascontiguousarray(color)
        kind = 1 if color.dtype == np.uint8 else 2
    E = np.eye(4) if extrinsic is None else np.asarray(extrinsic, np.float64).reshape(4, 4)
    pose = np.ascontiguousarray(np.linalg.inv(E).astype(np.float32).T)   # column-major
    K = _f32(np.asarray(intrinsic4, np.float32))
    count = (w // stride) * (h // stride)
    op = np.empty((count, 3), np.float32)
    on = np.empty((count, 3), np.float32) if compute_normals else None
    oc = np.empty((count, 3), np.float32) if color is not None else None
    m = int(lib().oracle_create_from_depth(
        _p(depth), None if color is None else color.ctypes.data_as(C.c_void_p), C.c_int(kind),
        C.c_int(w), C.c_int(h), _p(K), _p(pose), C.c_float(depth_cutoff), C.c_int(stride),
        C.c_int(int(rgbd)), C.c_int(int(compute_normals)), C.c_int(int(valid_only)),
        _p(op), _p(on), _p(oc)))
    return op[:m].copy(), (None if on is None else on[:m].copy()), (None if oc is None else oc[:m].copy())


def kinfu_pose_estimation(extrinsic, frame_pyramid, model_pyramid, distance_threshold=0.5,
                          icp_iterations=(20, 20, 20, 20), colored=False):
    """KinfuPipeline::PoseEstimation (kinfu/kinfu.cpp:105-143): coarse-to-fine RegistrationICP,
    point-to-plane with det_thresh 100000 (or Colored ICP, lambda 0.968), every level started
    from the level above's result.  Pyramids: lists of dicts {points, normals[, colors]},
    level 0 = finest."""
    T = np.asarray(extrinsic, np.float32).reshape(4, 4)
    for level in range(len(frame_pyramid) - 1, -1, -1):
        f, g = frame_pyramid[level], model_pyramid[level]
        if colored:
            res = registration_colored_icp(f["points"], g["points"], distance_threshold, f["colors"],
                                           g["colors"], g["normals"], init=T, lambda_geometric=0.968,
                                           det_thresh=100000.0, max_iteration=icp_iterations[level])
        else:
            res = registration_icp(f["points"], g["points"], distance_threshold, init=T, est=EST_PT2PL,
                                   det_thresh=100000.0, tgt_nrm=g["normals"],
                                   max_iteration=icp_iterations[level])
        T = res.transformation
    return T


def estimate_normals_radius(pts, radius, max_nn=30):
    pts = _f32(pts, (-1, 3))
    out = np.empty_like(pts)
    lib().oracle_estimate_normals_radius(_p(pts), C.c_int64(len(pts)), C.c_float(radius),
                                         C.c_int(max_nn), _p(out))
    return out


# ---------------------------------------------------------------------------
# RGB-D odometry (oracle/odometry_oracle.c)
# ---------------------------------------------------------------------------
OD_COLOR_TERM, OD_HYBRID_TERM = 0, 1


def od_filter(img, kind):
    """Image::Filter: kind 0 Gaussian3, 1 Sobel3Dx, 2 Sobel3Dy"""
    img = _f32(img)
    h, w = img.shape
    out = np.empty_like(img)
    lib().oracle_od_filter(_p(img), C.c_int(w), C.c_int(h), C.c_int(kind), _p(out))
    return out


def od_downsample(img):
    img = _f32(img)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), np.float32)
    lib().oracle_od_downsample(_p(img), C.c_int(w), C.c_int(h), _p(out))
    return out


def od_correspondence(K, extrinsic, depth_s, depth_t, max_depth_diff):
    depth_s, depth_t = _f32(depth_s), _f32(depth_t)
    h, w = depth_s.shape
    K = _f32(np.asarray(K, np.float32).reshape(3, 3))
    out = np.empty((h * w, 4), np.int32)
    n = lib().oracle_od_correspondence(_p(K), _p(_T_in(extrinsic)), _p(depth_s), _p(depth_t), C.c_int(w), C.c_int(h),
                                       C.c_float(max_depth_diff), out.ctypes.data_as(C.c_void_p))
    return out[:int(n)].copy()


def od_jacobian(hybrid, row, corr, source_color, target_color, target_depth, source_xyz, dx_color, dx_depth,
                dy_color, dy_depth, K, extrinsic):
    """RGBDOdometryJacobianFrom{Color,Hybrid}Term::ComputeJacobianAndResidual"""
    w = np.asarray(source_color).shape[1]
    corr = np.ascontiguousarray(corr, np.int32)
    J0, J1 = np.zeros(6, np.float32), np.zeros(6, np.float32)
    r0, r1 = C.c_float(0), C.c_float(0)
    f = lambda a: _p(_f32(a))
    K = _f32(np.asarray(K, np.float32).reshape(3, 3))
    lib().oracle_od_jacobian(C.c_int(int(hybrid)), C.c_int(row), corr.ctypes.data_as(C.c_void_p), f(source_color),
                             f(target_color), f(target_depth), f(source_xyz), f(dx_color), f(dx_depth), f(dy_color),
                             f(dy_depth), C.c_int(w), _p(K), _p(_T_in(extrinsic)), _p(J0), C.byref(r0), _p(J1),
                             C.byref(r1))
    return J0, float(r0.value), J1, float(r1.value)


def compute_rgbd_odometry(src_color, src_depth, tgt_color, tgt_depth, intrinsic4, odo_init=None,
                          jacobian=OD_HYBRID_TERM, iterations=(20, 10, 5), max_depth_diff=0.03, min_depth=0.0,
                          max_depth=4.0):
    """odometry::ComputeRGBDOdometry -> (success, 4x4 transformation, 6x6 information)"""
    sc, sd, tc, td = _f32(src_color), _f32(src_depth), _f32(tgt_color), _f32(tgt_depth)
    h, w = sc.shape
    init = np.eye(4, dtype=np.float32) if odo_init is None else odo_init
    it = (C.c_int * len(iterations))(*[int(v) for v in iterations])
    T = np.empty(16, np.float32)
    info = np.empty(36, np.float64)
    ok = lib().oracle_od_compute(_p(sc), _p(sd), _p(tc), _p(td), C.c_int(w), C.c_int(h),
                                 _p(_f32(np.asarray(intrinsic4, np.float32))), _p(_T_in(init)), C.c_int(int(jacobian)),
                                 it, C.c_int(len(iterations)), C.c_float(max_depth_diff), C.c_float(min_depth),
                                 C.c_float(max_depth), _p(T), info.ctypes.data_as(C.c_void_p))
    return bool(ok), T.reshape(4, 4).T.copy(), info.reshape(6, 6).copy()


def compute_weighted_rgbd_odometry(src_color, src_depth, tgt_color, tgt_depth, intrinsic4, odo_init=None,
                                   prev_twist=None, iterations=(20, 10, 5), max_depth_diff=0.03, min_depth=0.0,
                                   max_depth=4.0, nu=5.0, sigma2_init=1.0, inv_sigma_mat_diag=None):
    """odometry::ComputeWeightedRGBDOdometry -> (success, transformation, twist (6), information)"""
    sc, sd, tc, td = _f32(src_color), _f32(src_depth), _f32(tgt_color), _f32(tgt_depth)
    h, w = sc.shape
    init = np.eye(4, dtype=np.float32) if odo_init is None else odo_init
    pt = _f32(np.zeros(6) if prev_twist is None else prev_twist)
    isd = _f32(np.zeros(6) if inv_sigma_mat_diag is None else inv_sigma_mat_diag)
    it = (C.c_int * len(iterations))(*[int(v) for v in iterations])
    T, tw, info = np.empty(16, np.float32), np.empty(6, np.float32), np.empty(36, np.float64)
    ok = lib().oracle_od_compute_weighted(
        _p(sc), _p(sd), _p(tc), _p(td), C.c_int(w), C.c_int(h), _p(_f32(np.asarray(intrinsic4, np.float32))),
        _p(_T_in(init)), _p(pt), it, C.c_int(len(iterations)), C.c_float(max_depth_diff), C.c_float(min_depth),
        C.c_float(max_depth), C.c_float(nu), C.c_float(sigma2_init), _p(isd), _p(T), _p(tw),
        info.ctypes.data_as(C.c_void_p))
    return bool(ok), T.reshape(4, 4).T.copy(), tw.copy(), info.reshape(6, 6).copy()


def matrix4_to_vector6(T):
    """utility::TransformMatrix4fToVector6f"""
    out = np.empty(6, np.float32)
    lib().oracle_matrix4_to_vector6(_p(_T_in(T)), _p(out))
    return out


# ---------------------------------------------------------------------------
# oracle/_ref : the reference's own code compiled where it lies (optional)
# ---------------------------------------------------------------------------
def _ref_lib(name):
    path = os.path.join(_HERE, "_ref", name)
    return C.CDLL(path) if os.path.exists(path) else None


def ref_rand_vec3f(n, vmin, vmax, seed):
    """unit_test::Rand(Vector3f) driven by the reference's Raw generator."""
    L = _ref_lib("libref_raw.so")
    if L is None:
        raise FileNotFoundError("oracle/_ref/libref_raw.so not built (needs /root/reference)")
    out = np.empty((n, 3), np.float32)
    vmin, vmax = _f32(vmin, (3,)), _f32(vmax, (3,))
    L.ref_rand_vec3f(_p(out), C.c_int(n), _p(vmin), _p(vmax), C.c_int(seed))
    return out


def ref_rand_floats(n, vmin, vmax, seed):
    """unit_test::Rand(float*, n, vmin, vmax, seed) driven by the reference's Raw generator."""
    L = _ref_lib("libref_raw.so")
    if L is None:
        raise FileNotFoundError("oracle/_ref/libref_raw.so not built (needs /root/reference)")
    out = np.empty(n, np.float32)
    L.ref_rand_floats(_p(out), C.c_int(n), C.c_float(vmin), C.c_float(vmax), C.c_int(seed))
    return out


def ref_rand_vec4i(n, vmin, vmax, seed):
    L = _ref_lib("libref_raw.so")
    if L is None:
        raise FileNotFoundError("oracle/_ref/libref_raw.so not built (needs /root/reference)")
    out = np.empty((n, 4), np.int32)
    L.ref_rand_vec4i(out.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_int(vmin), C.c_int(vmax), C.c_int(seed))
    return out


def ref_lzf():
    """The reference's vendored liblzf (third_party/liblzf/lzf_c.c, lzf_d.c compiled in place into
    oracle/_ref/libref_lzf.so) or None where it was not built."""
    path = os.path.join(_HERE, "_ref", "libref_lzf.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.lzf_compress.restype = C.c_uint
    L.lzf_compress.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint]
    L.lzf_decompress.restype = C.c_uint
    L.lzf_decompress.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint]
    return L


def ref_flann_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_flann.so"))


def ref_flann_knn(tgt, qry, k):
    L = _ref_lib("libref_flann.so")
    tgt, qry = _f32(tgt, (-1, 3)), _f32(qry, (-1, 3))
    idx = np.empty((len(qry), k), np.int32)
    d2 = np.empty((len(qry), k), np.float32)
    L.ref_flann_knn(_p(tgt), C.c_int(len(tgt)), _p(qry), C.c_int(len(qry)), C.c_int(k),
                    _p(idx), _p(d2))
    return idx, d2


def ref_flann_radius(tgt, qry, radius, max_nn):
    L = _ref_lib("libref_flann.so")
    tgt, qry = _f32(tgt, (-1, 3)), _f32(qry, (-1, 3))
    idx = np.empty((len(qry), max_nn), np.int32)
    d2 = np.empty((len(qry), max_nn), np.float32)
    r = L.ref_flann_radius(_p(tgt), C.c_int(len(tgt)), _p(qry), C.c_int(len(qry)),
                           C.c_float(radius), C.c_int(max_nn), _p(idx), _p(d2))
    return int(r), idx, d2


def bench_iteration(src, tgt, tgt_nrm, max_dist, n_sample, repeats=1, n_single=0):
    """(build_s, iter_s, fitness[, iter_single_s]): wall-clock of one point-to-plane iteration
    over the first n_sample source points against the full target (bench.py cpu_baseline);
    n_single > 0 also times it on one thread over the first n_single points."""
    src, tgt, tn = _f32(src, (-1, 3)), _f32(tgt, (-1, 3)), _f32(tgt_nrm, (-1, 3))
    b, t, f, t1 = C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
    lib().oracle_bench_iteration(_p(src), C.c_int64(len(src)), _p(tgt), _p(tn),
                                 C.c_int64(len(tgt)), C.c_float(max_dist), C.c_int64(n_sample),
                                 C.c_int(repeats), C.byref(b), C.byref(t), C.byref(f),
                                 C.c_int64(n_single), C.byref(t1))
    if n_single > 0:
        return b.value, t.value, f.value, t1.value
    return b.value, t.value, f.value
