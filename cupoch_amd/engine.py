"""Thin object wrapper over the C ABI (include/mi_icp.h).

Inputs may be numpy arrays (host memory, copied by the engine) or torch CUDA
tensors (read in place on the device).  4x4 transforms are row-major numpy at
this level (what a user writes); the C ABI takes Eigen's column-major layout,
so they are transposed on the way in and out.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MI_ICP_DEVICE, MI_ICP_HOST, MiIcpError, Params, Result

try:  # torch is plumbing (device memory, streams), not a hard requirement of the host path
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_tensor(a):
    return torch is not None and isinstance(a, torch.Tensor)


class _Buf:
    """A float32/int32 array argument: keeps the backing object alive and exposes
    (pointer, mem_kind)."""

    def __init__(self, a, dtype, cols, device_index, copy=False):
        self.keep = None
        self.ptr = None
        self.kind = MI_ICP_HOST
        self.n = 0
        if a is None:
            return
        if _is_tensor(a):
            tdt = {np.float32: torch.float32, np.int32: torch.int32}[dtype]
            t = a
            if t.dtype != tdt:
                t = t.to(tdt)
            t = t.reshape(-1, cols).contiguous()
            if t.is_cuda:
                if t.device.index != device_index:
                    raise MiIcpError("tensor on cuda:%s passed to an engine on cuda:%s"
                                     % (t.device.index, device_index))
                self.kind = MI_ICP_DEVICE
                self.ptr = C.c_void_p(t.data_ptr())
            else:
                t = t.numpy()
                self.ptr = t.ctypes.data_as(C.c_void_p)
            self.keep = t
            self.n = int(t.shape[0])
        else:
            arr = np.ascontiguousarray(np.asarray(a, dtype=dtype).reshape(-1, cols))
            if copy and np.shares_memory(arr, a):
                arr = arr.copy()
            self.keep = arr
            self.ptr = arr.ctypes.data_as(C.c_void_p)
            self.n = int(arr.shape[0])


def _T_in(T):
    if T is None:
        return None, None
    if _is_tensor(T):
        T = T.detach().cpu().numpy()
    a = np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(4, 4).T)
    return a, a.ctypes.data_as(C.c_void_p)


def _T_out(buf16):
    return np.array(buf16, dtype=np.float32).reshape(4, 4).T.copy()


class Engine:
    """One mi_icp context = one GPU."""

    def __init__(self, device=0, use_torch_stream=True):
        self._L = _lib.load()
        self._ctx = C.c_void_p()
        rc = self._L.mi_icp_create(int(device), C.byref(self._ctx))
        if rc != 0:
            self._ctx = None
            raise MiIcpError("mi_icp_create(device=%d) failed with status %d "
                             "(no MI355X visible?)" % (device, rc))
        self.device = int(device)
        self.n_source = 0
        self.n_target = 0
        if use_torch_stream and torch is not None and torch.cuda.is_available():
            with torch.cuda.device(self.device):
                self.set_stream(torch.cuda.current_stream().cuda_stream)

    # -- plumbing --------------------------------------------------------------
    def _chk(self, rc):
        if rc < 0:
            msg = self._L.mi_icp_last_error(self._ctx)
            raise MiIcpError("mi_icp error %d: %s" % (rc, (msg or b"").decode()))
        return rc

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.mi_icp_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        self._chk(self._L.mi_icp_set_stream(self._ctx, C.c_void_p(int(hip_stream))))

    def synchronize(self):
        self._chk(self._L.mi_icp_synchronize(self._ctx))

    def _same_kind(self, *bufs):
        kinds = {b.kind for b in bufs if b.ptr is not None}
        if len(kinds) > 1:
            raise MiIcpError("all arrays of one call must live on the same side (host or device)")
        return kinds.pop() if kinds else MI_ICP_HOST

    # -- clouds ------------------------------------------------------------------
    def set_target(self, points, normals=None, covariances=None):
        self.generation = getattr(self, "generation", 0) + 1   # the clouds changed (registration._generic_icp)
        p = _Buf(points, np.float32, 3, self.device)
        n = _Buf(normals, np.float32, 3, self.device)
        c = _Buf(_cov_in(covariances), np.float32, 9, self.device)
        kind = self._same_kind(p, n, c)
        self._chk(self._L.mi_icp_set_target(self._ctx, p.ptr, n.ptr, c.ptr, p.n, kind))
        self.synchronize()  # the staging copies above may be freed by the caller now
        self.n_target = p.n

    def set_source(self, points, normals=None, covariances=None):
        self.generation = getattr(self, "generation", 0) + 1
        p = _Buf(points, np.float32, 3, self.device)
        n = _Buf(normals, np.float32, 3, self.device)
        c = _Buf(_cov_in(covariances), np.float32, 9, self.device)
        kind = self._same_kind(p, n, c)
        self._chk(self._L.mi_icp_set_source(self._ctx, p.ptr, n.ptr, c.ptr, p.n, kind))
        self.synchronize()
        self.n_source = p.n

    # -- KDTreeFlann-style search against the target ---------------------------------------
    def search_knn(self, queries, knn, radius=0.0):
        """(found, idx[nq, knn] int32, d2[nq, knn] float32): the knn nearest target points of
        every query (within `radius` when > 0), ascending; -1 / +inf padding.  Replaces the
        context's source cloud."""
        q = _Buf(queries, np.float32, 3, self.device)
        knn = int(knn)
        if q.kind == MI_ICP_DEVICE:
            idx = torch.empty((q.n, knn), dtype=torch.int32, device=q.keep.device)
            d2 = torch.empty((q.n, knn), dtype=torch.float32, device=q.keep.device)
            ip, dp = C.c_void_p(idx.data_ptr()), C.c_void_p(d2.data_ptr())
        else:
            idx = np.empty((q.n, knn), np.int32)
            d2 = np.empty((q.n, knn), np.float32)
            ip, dp = idx.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p)
        found = C.c_int64(0)
        self._chk(self._L.mi_icp_search_knn(self._ctx, q.ptr, q.n, knn, float(radius), ip, dp,
                                            C.byref(found), q.kind))
        self.n_source = q.n
        self.generation = getattr(self, "generation", 0) + 1
        return int(found.value), idx, d2

    # -- colored ICP --------------------------------------------------------------------
    def set_target_colors(self, colors):
        b = _Buf(colors, np.float32, 3, self.device)
        if b.n and b.n != self.n_target:
            raise MiIcpError("set_target_colors: %d colours for %d target points" % (b.n, self.n_target))
        self._chk(self._L.mi_icp_set_target_colors(self._ctx, b.ptr, b.kind))
        self.synchronize()

    def set_source_colors(self, colors):
        b = _Buf(colors, np.float32, 3, self.device)
        if b.n and b.n != self.n_source:
            raise MiIcpError("set_source_colors: %d colours for %d source points" % (b.n, self.n_source))
        self._chk(self._L.mi_icp_set_source_colors(self._ctx, b.ptr, b.kind))
        self.synchronize()

    def set_lambda_geometric(self, lambda_geometric):
        self._chk(self._L.mi_icp_set_lambda_geometric(self._ctx, float(lambda_geometric)))

    def compute_color_gradients(self, radius, max_nn=30, want_output=True):
        """InitializePointCloudForColoredICP; returns the gradients (target's original
        order, numpy) when want_output."""
        out, optr = None, None
        if want_output:
            out = np.empty((self.n_target, 3), np.float32)
            optr = out.ctypes.data_as(C.c_void_p)
        self._chk(self._L.mi_icp_compute_color_gradients(self._ctx, float(radius), int(max_nn), optr,
                                                         MI_ICP_HOST))
        return out

    def registration_colored_icp(self, max_distance, init=None, relative_fitness=1e-6,
                                 relative_rmse=1e-6, max_iteration=30, lambda_geometric=0.968,
                                 det_thresh=1e-6):
        res = Result()
        prm = Params(float(relative_fitness), float(relative_rmse), int(max_iteration),
                     float(det_thresh))
        _, tp = _T_in(init)
        self._chk(self._L.mi_icp_registration_colored_icp(self._ctx, float(max_distance), tp,
                                                          C.byref(prm), float(lambda_geometric),
                                                          C.byref(res)))
        return res

    def morton_order(self, points):
        """order[s] = original index of the s-th point of the engine's spatial (Morton) order;
        numpy int64.  points: numpy or a torch tensor on the engine's device (sorted there)."""
        p = _Buf(points, np.float32, 3, self.device)
        if p.n == 0:
            return np.zeros(0, np.int64)
        if p.kind == MI_ICP_DEVICE:
            out = torch.empty(p.n, dtype=torch.int32, device=p.keep.device)
            self._chk(self._L.mi_icp_spatial_order(self._ctx, p.ptr, p.n, C.c_void_p(out.data_ptr()), p.kind))
            return out.cpu().numpy().view(np.uint32).astype(np.int64)
        out = np.empty(p.n, np.uint32)
        self._chk(self._L.mi_icp_spatial_order(self._ctx, p.ptr, p.n, out.ctypes.data_as(C.c_void_p), p.kind))
        return out.astype(np.int64)

    def set_global_source_count(self, n_total):
        self._chk(self._L.mi_icp_set_global_source_count(self._ctx, int(n_total)))

    # -- search ---------------------------------------------------------------------
    def search_radius_1nn(self, radius, T=None, want_d2=True):
        """(indices[int32 n], d2[float32 n], stats) in original source order;
        -1 / +inf where no target point lies within `radius` (strict)."""
        idx = np.empty(self.n_source, np.int32)
        d2 = np.empty(self.n_source, np.float32) if want_d2 else None
        stats = np.zeros(3, np.float64)
        _, tp = _T_in(T)
        self._chk(self._L.mi_icp_search_radius_1nn(
            self._ctx, tp, float(radius), idx.ctypes.data_as(C.c_void_p),
            None if d2 is None else d2.ctypes.data_as(C.c_void_p), MI_ICP_HOST,
            stats.ctypes.data_as(C.c_void_p)))
        return idx, d2, stats

    def drop_seeds(self):
        """test hook (mi_icp_debug.h): the next search starts top-down, not from the previous matches"""
        self._chk(self._L.mi_icp_debug_drop_seeds(self._ctx))

    def last_search_kind(self):
        """test hook: 0 = the last search started at the root, 1 = from the previous matches, 2 = from its own seeds"""
        return int(self._L.mi_icp_debug_last_search_kind(self._ctx))

    def get_correspondences(self):
        cnt = C.c_int64(0)
        self._chk(self._L.mi_icp_get_correspondences(self._ctx, None, 0, C.byref(cnt), MI_ICP_HOST))
        out = np.empty((max(cnt.value, 0), 2), np.int32)
        if cnt.value > 0:
            self._chk(self._L.mi_icp_get_correspondences(
                self._ctx, out.ctypes.data_as(C.c_void_p), cnt.value, C.byref(cnt), MI_ICP_HOST))
        return out

    def set_correspondences(self, pairs):
        b = _Buf(pairs, np.int32, 2, self.device)
        self._chk(self._L.mi_icp_set_correspondences(self._ctx, b.ptr, b.n, b.kind))

    # -- estimation -------------------------------------------------------------------
    def compute_system(self, est, T=None):
        out = np.zeros(32, np.float64)
        _, tp = _T_in(T)
        self._chk(self._L.mi_icp_compute_system(self._ctx, int(est), tp,
                                                out.ctypes.data_as(C.c_void_p)))
        return out

    def compute_transformation(self, est, T=None, det_thresh=1e-6):
        out = (C.c_float * 16)()
        _, tp = _T_in(T)
        self._chk(self._L.mi_icp_compute_transformation(self._ctx, int(est), tp,
                                                        float(det_thresh), out))
        return _T_out(out)

    def compute_rmse(self, est, T=None):
        out = C.c_float(0)
        _, tp = _T_in(T)
        self._chk(self._L.mi_icp_compute_rmse(self._ctx, int(est), tp, C.byref(out)))
        return float(out.value)

    # -- registration --------------------------------------------------------------------
    def evaluate_registration(self, max_distance, T=None):
        res = Result()
        _, tp = _T_in(T)
        self._chk(self._L.mi_icp_evaluate_registration(self._ctx, float(max_distance), tp,
                                                       C.byref(res)))
        return res

    def registration_icp(self, est, max_distance, init=None, relative_fitness=1e-6,
                         relative_rmse=1e-6, max_iteration=30, det_thresh=1e-6):
        res = Result()
        prm = Params(float(relative_fitness), float(relative_rmse), int(max_iteration),
                     float(det_thresh))
        _, tp = _T_in(init)
        self._chk(self._L.mi_icp_registration_icp(self._ctx, int(est), float(max_distance), tp,
                                                  C.byref(prm), C.byref(res)))
        return res

    def icp_begin(self, est, max_distance, init=None, det_thresh=1e-6):
        res = Result()
        _, tp = _T_in(init)
        self._chk(self._L.mi_icp_icp_begin(self._ctx, int(est), float(max_distance), tp,
                                           float(det_thresh), C.byref(res)))
        return res

    def icp_iterate(self, n_iterations=1):
        res = Result()
        self._chk(self._L.mi_icp_icp_iterate(self._ctx, int(n_iterations), C.byref(res)))
        return res

    # -- geometry ---------------------------------------------------------------------------
    def transform(self, T, points=None, normals=None, covariances=None):
        """In place on torch CUDA tensors (as PointCloud::Transform); numpy inputs are
        left untouched and transformed copies are returned."""
        p = _Buf(points, np.float32, 3, self.device, copy=True)
        n = _Buf(normals, np.float32, 3, self.device, copy=True)
        c = _Buf(_cov_in(covariances), np.float32, 9, self.device, copy=True)
        kind = self._same_kind(p, n, c)
        cnt = max(p.n, n.n, c.n)
        _, tp = _T_in(T)
        self._chk(self._L.mi_icp_transform(self._ctx, tp, p.ptr, n.ptr, c.ptr, cnt, kind))
        return p.keep, n.keep, _cov_out(c.keep)

    def compute_bounds(self, points):
        """(min_bound, max_bound, center) of a cloud as float32 numpy vectors
        (GeometryBase3D::GetMinBound / GetMaxBound / GetCenter); zeros for an empty cloud."""
        p = _Buf(points, np.float32, 3, self.device)
        out = np.zeros((3, 3), np.float32)
        ptr = lambda r: out[r].ctypes.data_as(C.c_void_p)
        self._chk(self._L.mi_icp_compute_bounds(self._ctx, p.ptr, p.n, p.kind, ptr(0), ptr(1), ptr(2)))
        return out[0].copy(), out[1].copy(), out[2].copy()

    def affine(self, points=None, normals=None, covariances=None, R=None, scale=None, center=None, translate=None):
        """GeometryBase3D::Translate / Scale / Rotate: p <- (R (p - center)) * scale + center + translate,
        normals <- R n, covariances <- R C R^T.  In place on torch CUDA tensors; numpy inputs are left
        untouched and moved copies are returned, as Engine.transform."""
        p = _Buf(points, np.float32, 3, self.device, copy=True)
        n = _Buf(normals, np.float32, 3, self.device, copy=True)
        c = _Buf(_cov_in(covariances), np.float32, 9, self.device, copy=True)
        kind = self._same_kind(p, n, c)
        cnt = max(p.n, n.n, c.n)
        vec = lambda v: None if v is None else np.ascontiguousarray(np.asarray(v, np.float32).reshape(3))
        Rc = None if R is None else np.ascontiguousarray(np.asarray(R, np.float32).reshape(3, 3).T)   # column-major
        cv, tv = vec(center), vec(translate)
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        self._chk(self._L.mi_icp_affine(self._ctx, ptr(Rc), float(scale if scale is not None else 1.0),
                                        0 if scale is None else 1, ptr(cv), ptr(tv), p.ptr, n.ptr, c.ptr, cnt, kind))
        self.synchronize()
        return p.keep, n.keep, _cov_out(c.keep)

    def voxel_downsample(self, points, voxel_size, normals=None, colors=None):
        p = _Buf(points, np.float32, 3, self.device)
        n = _Buf(normals, np.float32, 3, self.device)
        c = _Buf(colors, np.float32, 3, self.device)
        kind = self._same_kind(p, n, c)
        m = C.c_int64(0)
        if kind == MI_ICP_DEVICE:
            dev = p.keep.device
            mk = lambda b: torch.empty((p.n, 3), dtype=torch.float32, device=dev) if b.ptr is not None else None
            ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        else:
            mk = lambda b: np.empty((p.n, 3), np.float32) if b.ptr is not None else None
            ptr = lambda t: None if t is None else t.ctypes.data_as(C.c_void_p)
        op, on, oc = mk(p), mk(n), mk(c)
        self._chk(self._L.mi_icp_voxel_downsample(self._ctx, p.ptr, n.ptr, c.ptr, p.n,
                                                  float(voxel_size), ptr(op), ptr(on), ptr(oc),
                                                  C.byref(m), kind))
        k = int(m.value)
        cut = lambda t: None if t is None else t[:k]
        return cut(op) if op is not None else np.empty((0, 3), np.float32), cut(on), cut(oc)

    def create_from_depth(self, depth, intrinsic4, extrinsic=None, color=None, depth_scale=1000.0,
                          depth_trunc=1000.0, depth_cutoff=-1.0, stride=1, rgbd=False,
                          compute_normals=False, valid_only=True):
        """PointCloud::CreateFromDepthImage (rgbd=False) / CreateFromRGBDImage (rgbd=True),
        geometry/pointcloud_factory.cu:286-376.  depth: [H, W] float32 or uint16; color: None,
        [H, W, 3] uint8 or [H, W] float32; numpy or torch (all on the same side).
        Returns (points, normals or None, colors or None)."""
        on_dev = _is_tensor(depth) and depth.is_cuda
        def prep(x, kinds):
            if x is None:
                return None, None
            if _is_tensor(x):
                if x.dtype not in kinds:
                    raise TypeError("unsupported image dtype %s" % x.dtype)
                if x.is_cuda != on_dev:
                    raise ValueError("depth and color must live on the same side")
                x = x.contiguous() if on_dev else np.ascontiguousarray(x.numpy())
            else:
                if on_dev:
                    raise ValueError("depth and color must live on the same side")
                x = np.ascontiguousarray(x)
            return x, (C.c_void_p(x.data_ptr()) if on_dev else x.ctypes.data_as(C.c_void_p))
        if _is_tensor(depth):
            dkinds, ckinds = (torch.float32, torch.uint16), (torch.uint8, torch.float32)
        else:
            dkinds = ckinds = None
        d, dptr = prep(depth, dkinds)
        col, cptr = prep(color, ckinds)
        if d.ndim != 2:
            raise ValueError("depth must be [H, W]")
        dname = str(d.dtype).replace("torch.", "")
        if dname not in ("float32", "uint16"):
            raise TypeError("depth must be float32 or uint16")
        h, w = int(d.shape[0]), int(d.shape[1])
        ctype = 0
        if col is not None:
            cname = str(col.dtype).replace("torch.", "")
            if cname == "uint8" and tuple(col.shape) == (h, w, 3):
                ctype = 1
            elif cname == "float32" and tuple(col.shape)[:2] == (h, w) and \
                    (col.numel() if on_dev else col.size) == h * w:
                ctype = 2
            else:
                raise TypeError("[PointCloud::CreateFromRGBDImage] Unsupported image format.")
        K = (C.c_float * 4)(*[float(v) for v in intrinsic4])
        E = None
        if extrinsic is not None:
            E = np.ascontiguousarray(np.asarray(extrinsic, np.float32).reshape(4, 4).T)
            Eptr = E.ctypes.data_as(C.c_void_p)
        else:
            Eptr = None
        count = (w // int(stride)) * (h // int(stride)) if stride >= 1 else 0
        kind = MI_ICP_DEVICE if on_dev else MI_ICP_HOST
        if on_dev:
            mk = lambda want: torch.empty((count, 3), dtype=torch.float32, device=d.device) if want else None
            ptr = lambda t_: None if t_ is None else C.c_void_p(t_.data_ptr())
        else:
            mk = lambda want: np.empty((count, 3), np.float32) if want else None
            ptr = lambda t_: None if t_ is None else t_.ctypes.data_as(C.c_void_p)
        op, on, oc = mk(True), mk(bool(compute_normals)), mk(col is not None)
        m = C.c_int64(0)
        self._chk(self._L.mi_icp_create_from_depth(
            self._ctx, dptr, 1 if dname == "uint16" else 0, cptr, ctype, w, h, K, Eptr,
            float(depth_scale), float(depth_trunc), float(depth_cutoff), int(stride), int(bool(rgbd)),
            int(bool(compute_normals)), int(bool(valid_only)), ptr(op), ptr(on), ptr(oc), C.byref(m), kind))
        k = int(m.value)
        cut = lambda t_: None if t_ is None else t_[:k]
        return cut(op), cut(on), cut(oc)

    def compute_rgbd_odometry(self, source_color, source_depth, target_color, target_depth, intrinsic4,
                              odo_init=None, jacobian=1, iterations=(20, 10, 5), max_depth_diff=0.03,
                              min_depth=0.0, max_depth=4.0, weighted=False, prev_twist=None, nu=5.0,
                              sigma2_init=1.0, inv_sigma_mat_diag=None):
        """odometry::ComputeRGBDOdometry / ComputeWeightedRGBDOdometry (odometry/odometry.cu:833-943).
        Images: [H, W] float32, numpy or torch (all on the same side).  Returns (success, 4x4
        transformation, 6x6 information), with weighted=True (success, transformation, twist, information)."""
        imgs = [source_color, source_depth, target_color, target_depth]
        on_dev = _is_tensor(imgs[0]) and imgs[0].is_cuda
        keep, ptrs = [], []
        for x in imgs:
            if _is_tensor(x):
                if x.is_cuda != on_dev or x.dtype != torch.float32:
                    raise TypeError("odometry images must be float32 and live on the same side")
                x = x.contiguous() if on_dev else np.ascontiguousarray(x.numpy())
            else:
                if on_dev:
                    raise TypeError("odometry images must live on the same side")
                x = np.ascontiguousarray(x)
                if x.dtype != np.float32:
                    raise TypeError("odometry images must be float32")
            keep.append(x)
            ptrs.append(C.c_void_p(x.data_ptr()) if on_dev else x.ctypes.data_as(C.c_void_p))
        shape = tuple(keep[0].shape)
        if len(shape) != 2 or any(tuple(k.shape) != shape for k in keep):
            raise ValueError("[RGBDOdometry] Two RGBD pairs should be same in size.")
        from ._lib import OdometryOption
        opt = OdometryOption()
        opt.num_levels = len(iterations)
        for i, v in enumerate(list(iterations)[:8]):   # (more than 8 levels: the library reports it)
            opt.iterations[i] = int(v)
        opt.max_depth_diff, opt.min_depth, opt.max_depth = float(max_depth_diff), float(min_depth), float(max_depth)
        opt.nu, opt.sigma2_init = float(nu), float(sigma2_init)
        for i in range(6):
            opt.inv_sigma_mat_diag[i] = 0.0 if inv_sigma_mat_diag is None else float(inv_sigma_mat_diag[i])
        K = (C.c_float * 4)(*[float(v) for v in intrinsic4])
        init = None
        if odo_init is not None:
            init = np.ascontiguousarray(np.asarray(odo_init, np.float32).reshape(4, 4).T)
        ok = C.c_int(0)
        T = np.empty(16, np.float32)
        info = np.empty(36, np.float64)
        if weighted:
            pt = (C.c_float * 6)(*([0.0] * 6 if prev_twist is None else [float(v) for v in prev_twist]))
            tw = np.empty(6, np.float32)
            self._chk(self._L.mi_icp_compute_weighted_rgbd_odometry(
                self._ctx, ptrs[0], ptrs[1], ptrs[2], ptrs[3], int(shape[1]), int(shape[0]), K,
                None if init is None else init.ctypes.data_as(C.c_void_p), pt, C.byref(opt), C.byref(ok),
                T.ctypes.data_as(C.c_void_p), tw.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p),
                MI_ICP_DEVICE if on_dev else MI_ICP_HOST))
            return bool(ok.value), T.reshape(4, 4).T.copy(), tw, info.reshape(6, 6).copy()
        self._chk(self._L.mi_icp_compute_rgbd_odometry(
            self._ctx, ptrs[0], ptrs[1], ptrs[2], ptrs[3], int(shape[1]), int(shape[0]), K,
            None if init is None else init.ctypes.data_as(C.c_void_p), int(jacobian), C.byref(opt), C.byref(ok),
            T.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p), MI_ICP_DEVICE if on_dev else MI_ICP_HOST))
        return bool(ok.value), T.reshape(4, 4).T.copy(), info.reshape(6, 6).copy()

    def covariances_from_normals(self, normals, epsilon=1e-3):
        n = _Buf(normals, np.float32, 3, self.device)
        if n.kind == MI_ICP_DEVICE:
            out = torch.empty((n.n, 9), dtype=torch.float32, device=n.keep.device)
            optr = C.c_void_p(out.data_ptr())
        else:
            out = np.empty((n.n, 9), np.float32)
            optr = out.ctypes.data_as(C.c_void_p)
        self._chk(self._L.mi_icp_covariances_from_normals(self._ctx, n.ptr, n.n, float(epsilon),
                                                          optr, n.kind))
        return _cov_out(out)

    def estimate_normals_knn(self, points, knn=30):
        p = _Buf(points, np.float32, 3, self.device)
        if p.kind == MI_ICP_DEVICE:
            out = torch.empty((p.n, 3), dtype=torch.float32, device=p.keep.device)
            optr = C.c_void_p(out.data_ptr())
        else:
            out = np.empty((p.n, 3), np.float32)
            optr = out.ctypes.data_as(C.c_void_p)
        self._chk(self._L.mi_icp_estimate_normals_knn(self._ctx, p.ptr, p.n, int(knn), optr, p.kind))
        return out

    def estimate_normals_radius(self, points, radius, max_nn=30):
        p = _Buf(points, np.float32, 3, self.device)
        if p.kind == MI_ICP_DEVICE:
            out = torch.empty((p.n, 3), dtype=torch.float32, device=p.keep.device)
            optr = C.c_void_p(out.data_ptr())
        else:
            out = np.empty((p.n, 3), np.float32)
            optr = out.ctypes.data_as(C.c_void_p)
        self._chk(self._L.mi_icp_estimate_normals_radius(self._ctx, p.ptr, p.n, float(radius),
                                                         int(max_nn), optr, p.kind))
        return out

    # -- multi-GPU / instrumentation -------------------------------------------------------------
    def comm_init(self, unique_id, nranks, rank):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self._L.mi_icp_comm_init(self._ctx, buf, int(nranks), int(rank)))

    def comm_init_local(self, job_name, nranks, rank):
        """node-local communicator: the shared-memory mailbox alone (csrc/mailbox.h), no RCCL"""
        self._chk(self._L.mi_icp_comm_init_local(self._ctx, str(job_name).encode(), int(nranks), int(rank)))

    def comm_kind(self):
        """0: none, 1: RCCL all-reduce, 2: shared-memory mailbox"""
        return int(self._L.mi_icp_comm_kind(self._ctx))

    def comm_autotune(self, exchanges=200):
        """Collective: self-test (known-answer) and time every available exchange path, keep the fastest that passed on
        every rank (mi_icp_comm_autotune).  Returns a dict: chosen ("host mailbox" / "device inboxes" / "rccl" / "none"),
        latency_us per path (None: not available, "failed": did not pass), rccl_comm_count, exchanges, verified."""
        lat = (C.c_double * 3)()
        info = (C.c_int * 4)()
        self._chk(self._L.mi_icp_comm_autotune(self._ctx, int(exchanges), lat, info))
        names = ("host mailbox", "device inboxes", "rccl")
        show = lambda v: None if v == -1.0 else ("failed" if v < 0 else round(float(v), 3))
        return {"chosen": names[info[0] - 1] if 1 <= info[0] <= 3 else "none",
                "latency_us": {n: show(lat[i]) for i, n in enumerate(names)},
                "rccl_comm_count": int(info[1]), "exchanges": int(info[2]), "verified": bool(info[3])}

    def comm_destroy(self):
        self._chk(self._L.mi_icp_comm_destroy(self._ctx))

    def set_iteration_callback(self, fn):
        """fn(iteration, fitness, inlier_rmse) once per iteration of the following loops, in order (what the
        reference logs at debug verbosity, registration.cu:155-156); None removes it."""
        from ._lib import ITERATION_FN
        self._iter_cb = None if fn is None else ITERATION_FN(lambda _user, i, f, r: fn(int(i), float(f), float(r)))
        self._chk(self._L.mi_icp_set_iteration_callback(self._ctx, C.cast(self._iter_cb, C.c_void_p) if self._iter_cb else None, None))

    def set_step_stamps(self, enable=True):
        """include/mi_icp_debug.h: the next loop runs the stamping instantiations of its kernels."""
        self._chk(self._L.mi_icp_debug_set_step_stamps(self._ctx, 1 if enable else 0))

    def get_step_stamps(self):
        """-> (32 stamp words as uint64, ticks per microsecond)"""
        out = np.zeros(32, np.uint64)
        tpu = C.c_double(0.0)
        self._chk(self._L.mi_icp_debug_get_step_stamps(self._ctx, out.ctypes.data_as(C.c_void_p), C.byref(tpu)))
        return out, float(tpu.value)

    def set_profiling(self, enable=True):
        self._chk(self._L.mi_icp_set_profiling(self._ctx, 1 if enable else 0))

    def get_profile(self):
        out = np.zeros(8, np.float64)
        self._chk(self._L.mi_icp_get_profile(self._ctx, out.ctypes.data_as(C.c_void_p)))
        return dict(nn_ms=out[0], nn_launches=int(out[1]), reduce_ms=out[2],
                    reduce_launches=int(out[3]), build_target_ms=out[4], build_source_ms=out[5],
                    halo_builds_by_loops=int(out[6]))


def comm_unique_id():
    L = _lib.load()
    buf = C.create_string_buffer(128)
    rc = L.mi_icp_comm_unique_id(buf)
    if rc != 0:
        raise MiIcpError("mi_icp_comm_unique_id failed (%d): RCCL not loadable" % rc)
    return bytes(buf.raw)


def _cov_in(covs):
    """(n,3,3) row-major user layout -> (n,9) column-major (Eigen::Matrix3f)."""
    if covs is None:
        return None
    if _is_tensor(covs):
        c = covs.reshape(-1, 3, 3)
        return c.transpose(1, 2).contiguous().reshape(-1, 9)
    c = np.asarray(covs, np.float32).reshape(-1, 3, 3)
    return np.ascontiguousarray(c.transpose(0, 2, 1)).reshape(-1, 9)


def _cov_out(c9):
    if c9 is None:
        return None
    if _is_tensor(c9):
        return c9.reshape(-1, 3, 3).transpose(1, 2).contiguous()
    return np.ascontiguousarray(np.asarray(c9).reshape(-1, 3, 3).transpose(0, 2, 1))


def solve_system(sys32, det_thresh=1e-6):
    L = _lib.load()
    sys32 = np.ascontiguousarray(sys32, np.float64)
    out = (C.c_float * 16)()
    ok = L.mi_icp_solve_system(sys32.ctypes.data_as(C.c_void_p), float(det_thresh), out)
    return bool(ok > 0), _T_out(out)


def kabsch_from_sums(sys32, n_model):
    L = _lib.load()
    sys32 = np.ascontiguousarray(sys32, np.float64)
    out = (C.c_float * 16)()
    L.mi_icp_kabsch_from_sums(sys32.ctypes.data_as(C.c_void_p), int(n_model), out)
    return _T_out(out)


def vector6_to_matrix4(x):
    L = _lib.load()
    x = np.ascontiguousarray(x, np.float32)
    out = (C.c_float * 16)()
    L.mi_icp_vector6_to_matrix4(x.ctypes.data_as(C.c_void_p), out)
    return _T_out(out)
