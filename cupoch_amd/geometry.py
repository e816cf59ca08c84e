"""cupoch.geometry.PointCloud mirror (src/cupoch/geometry/pointcloud.h:43-263,
python surface src/python/cupoch_pybind/geometry/pointcloud.cpp:33-200) --
only the members the ICP path touches.  Arrays live on the GPU as torch
tensors; every operation below runs a HIP kernel through the C ABI."""
import numpy as np

from . import utility
from .engine import Engine

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

_engines = {}


def get_engine(device=None):
    """Process-wide engine per GPU (the reference's per-process allocator/streams)."""
    d = utility.default_device() if device is None else int(device)
    if d not in _engines:
        _engines[d] = Engine(d)
    return _engines[d]


class KDTreeSearchParamKNN:
    """knn::KDTreeSearchParamKNN (knn/kdtree_search_param.h:49-56)"""

    def __init__(self, knn=30):
        self.knn = int(knn)


class KDTreeSearchParamRadius:
    """knn::KDTreeSearchParamRadius (knn/kdtree_search_param.h:58-66)"""

    def __init__(self, radius, max_nn):   # no default in the reference either
        self.radius = float(radius)
        self.max_nn = int(max_nn)


class KDTreeFlann:
    """knn::KDTreeFlann (knn/kdtree_flann.h:43-124; python surface
    cupoch_pybind/geometry/kdtree_flann.cpp:88-141).  Owns its own engine context, i.e. its
    own tree; at most knn::NUM_MAX_NN = 100 neighbours per query."""

    def __init__(self, geometry=None):
        self._eng = None
        self._n = 0
        if geometry is not None:
            self.set_geometry(geometry)

    def set_geometry(self, geometry):
        pts = geometry.points.tensor if isinstance(geometry, PointCloud) else _v3(geometry).tensor
        if self._eng is None:
            self._eng = Engine(pts.device.index if pts.is_cuda else utility.default_device())
        self._eng.set_target(pts)
        self._n = int(pts.shape[0])
        return True

    # batch forms: (found, idx[nq, k], d2[nq, k]) on the device
    def search_knn(self, queries, knn):
        if self._n == 0:
            return -1, None, None
        return self._eng.search_knn(_v3(queries).tensor if not isinstance(queries, utility.Vector3fVector)
                                    else queries.tensor, knn)

    def search_radius(self, queries, radius, max_nn):
        if self._n == 0 or not radius > 0.0:   # (a non-positive radius holds no neighbours; the engine reads 0 as "unbounded")
            return -1, None, None
        return self._eng.search_knn(_v3(queries).tensor if not isinstance(queries, utility.Vector3fVector)
                                    else queries.tensor, max_nn, radius)

    # single-query forms of the pybind module: (k, indices, distance2) as host lists of length k
    def search_knn_vector_3f(self, query, knn):
        if self._n == 0:
            raise RuntimeError("search_knn_vector_3f() error!")
        k, idx, d2 = self._eng.search_knn(np.asarray(query, np.float32).reshape(1, 3), knn)
        return k, list(idx[0, :k]), list(d2[0, :k])

    def search_radius_vector_3f(self, query, radius, max_nn):
        if self._n == 0:
            raise RuntimeError("search_radius_vector_3f() error!")
        if not radius > 0.0:
            return 0, [], []
        k, idx, d2 = self._eng.search_knn(np.asarray(query, np.float32).reshape(1, 3), max_nn, radius)
        return k, list(idx[0, :k]), list(d2[0, :k])

    def search_vector_3f(self, query, search_param):
        if isinstance(search_param, KDTreeSearchParamRadius):
            return self.search_radius_vector_3f(query, search_param.radius, search_param.max_nn)
        return self.search_knn_vector_3f(query, search_param.knn)


class AxisAlignedBoundingBox:
    """geometry::AxisAlignedBoundingBox<3> (geometry/boundingvolume.h:121-230) as far as the
    ICP path's PointCloud hands it out: bounds, extent, centre, volume."""

    def __init__(self, min_bound=(0, 0, 0), max_bound=(0, 0, 0)):
        self.min_bound = np.asarray(min_bound, np.float32).reshape(3).copy()
        self.max_bound = np.asarray(max_bound, np.float32).reshape(3).copy()
        self.color = np.zeros(3, np.float32)

    def get_min_bound(self):
        return self.min_bound

    def get_max_bound(self):
        return self.max_bound

    def get_center(self):
        return ((self.min_bound + self.max_bound) * np.float32(0.5)).astype(np.float32)

    def get_extent(self):
        return self.max_bound - self.min_bound

    def get_half_extent(self):
        return self.get_extent() * np.float32(0.5)

    def get_max_extent(self):
        return float(self.get_extent().max())

    def volume(self):
        return float(np.prod(self.get_extent()))

    def is_empty(self):
        return self.volume() <= 0

    def __repr__(self):
        return "geometry::AxisAlignedBoundingBox with min_bound %s and max_bound %s" % (self.min_bound, self.max_bound)


class PointCloud:
    def __init__(self, points=None):
        self._points = utility.Vector3fVector() if points is None else _v3(points)
        self._normals = None
        self._colors = None
        self._covariances = None

    # device-vector style properties ------------------------------------------------
    @property
    def points(self):
        return self._points

    @points.setter
    def points(self, v):
        self._points = _v3(v)

    @property
    def normals(self):
        return self._normals if self._normals is not None else utility.Vector3fVector()

    @normals.setter
    def normals(self, v):
        self._normals = _v3(v)

    @property
    def colors(self):
        return self._colors if self._colors is not None else utility.Vector3fVector()

    @colors.setter
    def colors(self, v):
        self._colors = _v3(v)

    @property
    def covariances(self):
        return self._covariances if self._covariances is not None else utility.Matrix3fVector()

    @covariances.setter
    def covariances(self, v):
        self._covariances = v if isinstance(v, utility.Matrix3fVector) else utility.Matrix3fVector(v)

    # pointcloud.h:82-96 ------------------------------------------------------------------
    def has_points(self):
        return len(self._points) > 0

    def has_normals(self):
        return self.has_points() and self._normals is not None and len(self._normals) == len(self._points)

    def has_colors(self):
        return self.has_points() and self._colors is not None and len(self._colors) == len(self._points)

    def has_covariances(self):
        return (self.has_points() and self._covariances is not None
                and len(self._covariances) == len(self._points))

    def is_empty(self):
        return not self.has_points()

    def clone(self):
        out = PointCloud()
        out._points = utility.Vector3fVector(self._points.tensor.clone())
        for name in ("_normals", "_colors"):
            v = getattr(self, name)
            if v is not None:
                setattr(out, name, utility.Vector3fVector(v.tensor.clone()))
        if self._covariances is not None:
            out._covariances = utility.Matrix3fVector(self._covariances.tensor.clone())
        return out

    # PointCloud::Transform (pointcloud.cu:293-299) ------------------------------------------
    # GeometryBase3D (geometry/geometry_base.h:44-90, geometry/pointcloud.cu:205-242) ---------------
    def _bounds(self):
        return get_engine(self._points.tensor.device.index if self._points.tensor.is_cuda else None) \
            .compute_bounds(self._points.tensor)

    def get_min_bound(self):
        return self._bounds()[0]

    def get_max_bound(self):
        return self._bounds()[1]

    def get_center(self):
        return self._bounds()[2]

    def get_axis_aligned_bounding_box(self):
        mn, mx, _ = self._bounds()
        return AxisAlignedBoundingBox(mn, mx)

    def _affine(self, with_attributes, **kw):
        eng = get_engine(self._points.tensor.device.index if self._points.tensor.is_cuda else None)
        n = self._normals.tensor if with_attributes and self._normals is not None and len(self._normals) else None
        c = self._covariances.tensor if with_attributes and self._covariances is not None and len(self._covariances) else None
        _, _, c_new = eng.affine(self._points.tensor, n, c, **kw)
        if c_new is not None:
            self._covariances.tensor = c_new
        return self

    def translate(self, translation, relative=True):
        t = np.asarray(translation, np.float32).reshape(3)
        if not relative:
            t = (t - self.get_center()).astype(np.float32)
        return self._affine(False, translate=t)

    def scale(self, scale, center=True):
        c = self.get_center() if center and len(self._points) else None
        return self._affine(False, scale=float(scale), center=c)

    def rotate(self, R, center=True):
        c = self.get_center() if center and len(self._points) else None
        return self._affine(True, R=np.asarray(R, np.float32).reshape(3, 3), center=c)

    def transform(self, transformation):
        eng = get_engine(self._points.tensor.device.index)
        n = self._normals.tensor if self._normals is not None and len(self._normals) else None
        c = self._covariances.tensor if self._covariances is not None and len(self._covariances) else None
        _, _, c_new = eng.transform(np.asarray(transformation, np.float32), self._points.tensor, n, c)
        if c_new is not None:
            self._covariances.tensor = c_new
        return self

    # PointCloud::VoxelDownSample (down_sample.cu:170-273) --------------------------------------
    def voxel_down_sample(self, voxel_size):
        out = PointCloud()
        if not self.has_points():
            return out
        eng = get_engine(self._points.tensor.device.index)
        n = self._normals.tensor if self.has_normals() else None
        c = self._colors.tensor if self.has_colors() else None
        p2, n2, c2 = eng.voxel_downsample(self._points.tensor, float(voxel_size), n, c)
        out._points = utility.Vector3fVector(p2.clone())
        if n2 is not None:
            out._normals = utility.Vector3fVector(n2.clone())
        if c2 is not None:
            out._colors = utility.Vector3fVector(c2.clone())
        return out

    # PointCloud::EstimateNormals (estimate_normals.cu:82-127): KNN or Radius search parameter ----------
    def estimate_normals(self, search_param=None):
        eng = get_engine(self._points.tensor.device.index)
        if isinstance(search_param, KDTreeSearchParamRadius):
            nrm = eng.estimate_normals_radius(self._points.tensor, search_param.radius, search_param.max_nn)
        else:
            k = 30 if search_param is None else int(getattr(search_param, "knn", 30))
            nrm = eng.estimate_normals_knn(self._points.tensor, k)
        self._normals = utility.Vector3fVector(nrm)
        return True


class Image:
    """geometry::Image as the factories below see it (geometry/image.h:52-110): a [H, W] or
    [H, W, C] array (numpy, or a torch tensor on either side); float32 / uint16 depth,
    uint8 x 3 or float32 x 1 colour.  Image processing (pyramids, filters) is out of scope."""

    def __init__(self, data=None):
        self.data = data

    @property
    def height(self):
        return 0 if self.data is None else int(self.data.shape[0])

    @property
    def width(self):
        return 0 if self.data is None else int(self.data.shape[1])


def _img(x):
    return x.data if isinstance(x, Image) else x


class RGBDImage:
    """geometry::RGBDImage (geometry/rgbdimage.h:38-120): color + float depth."""

    def __init__(self, color=None, depth=None):
        self.color = _img(color)
        self.depth = _img(depth)


def _create_from_depth_image(depth, intrinsic, extrinsic=None, depth_scale=1000.0, depth_trunc=1000.0, stride=1):
    """PointCloud::CreateFromDepthImage (pointcloud_factory.cu:329-351)"""
    d = _img(depth)
    name = str(d.dtype).replace("torch.", "")
    out = PointCloud()
    if d.ndim != 2 or name not in ("float32", "uint16"):
        print("[cupoch_amd] Error: [PointCloud::CreateFromDepthImage] Unsupported image format.")
        return out
    dev = d.device.index if (torch is not None and torch.is_tensor(d) and d.is_cuda) else None
    p, _, _ = get_engine(dev).create_from_depth(d, intrinsic.as4(), extrinsic, None, depth_scale, depth_trunc,
                                                -1.0, stride, False, False, True)
    out._points = utility.Vector3fVector(p)
    return out


def _create_from_rgbd_image(image, intrinsic, extrinsic=None, project_valid_depth_only=True, depth_cutoff=-1.0,
                            compute_normals=False):
    """PointCloud::CreateFromRGBDImage (pointcloud_factory.cu:353-376).  image.color may be
    None (depth-only frames, as KinFu's point-to-plane tracking uses them)."""
    out = PointCloud()
    d, c = image.depth, image.color
    if str(d.dtype).replace("torch.", "") != "float32":
        print("[cupoch_amd] Error: [PointCloud::CreateFromRGBDImage] Unsupported image format.")
        return out
    dev = d.device.index if (torch is not None and torch.is_tensor(d) and d.is_cuda) else None
    try:
        p, n, col = get_engine(dev).create_from_depth(d, intrinsic.as4(), extrinsic, c, 1000.0, 1000.0, depth_cutoff,
                                                      1, True, compute_normals, project_valid_depth_only)
    except TypeError:
        print("[cupoch_amd] Error: [PointCloud::CreateFromRGBDImage] Unsupported image format.")
        return out
    out._points = utility.Vector3fVector(p)
    if n is not None:
        out._normals = utility.Vector3fVector(n)
    if col is not None:
        out._colors = utility.Vector3fVector(col)
    return out


PointCloud.create_from_depth_image = staticmethod(_create_from_depth_image)
PointCloud.create_from_rgbd_image = staticmethod(_create_from_rgbd_image)


def _v3(v):
    return v if isinstance(v, utility.Vector3fVector) else utility.Vector3fVector(v)
