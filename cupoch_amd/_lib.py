"""Loader for libmi_icp.so (the HIP engine behind this package).

There is no CPU fallback: if the shared library is missing or cannot be
loaded, importing a compute entry point raises.  `build()` compiles it in-tree
with hipcc for gfx950 (it cross-compiles on a box without a GPU).
"""
import ctypes as C
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmi_icp.so")
INCLUDE = os.path.join(ROOT, "include")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I/opt/rocm/include"]
# the library's translation units (csrc/ctx.h says what each holds); compiled side by side, then linked
UNITS = ["mi_icp", "mi_build", "mi_geometry", "mi_knn", "mi_comm", "mi_debug"]
OBJ_DIR = os.path.join(LIB_DIR, "obj")

MI_ICP_HOST, MI_ICP_DEVICE = 0, 1
EST_POINT_TO_POINT, EST_POINT_TO_PLANE, EST_SYMMETRIC, EST_GENERALIZED = 1, 2, 3, 5
EST_COLORED = 4


class MiIcpError(RuntimeError):
    pass


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith((".hip", ".h"))) + [os.path.join(INCLUDE, "mi_icp.h"),
                                                 os.path.join(INCLUDE, "mi_icp_debug.h")]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in _sources())


def build(force=False, verbose=False, variant=None, defines=()):
    """hipcc --offload-arch=gfx950 -c csrc/<unit>.hip for every unit (in parallel; a unit whose object is newer than
    every source is kept), then hipcc -shared ... -o cupoch_amd/lib/libmi_icp.so
    variant / defines: a second build for same-box A/B runs (MI_ICP_LIB_PATH), e.g. variant="h4",
    defines=("-DMI_HALO_STORED=4",) -> cupoch_amd/lib/libmi_icp_h4.so
    Safe against several processes building at once (every rank of a multi-process test imports the package): one
    file lock around the whole build, objects and the library written under a temporary name and renamed; objects
    compiled with other flags / defines are never reused (a stamp in the object directory holds their hash)."""
    lib_path = LIB_PATH if not variant else os.path.join(LIB_DIR, "libmi_icp_%s.so" % variant)
    obj_dir = OBJ_DIR if not variant else OBJ_DIR + "_" + variant
    if not force and not variant and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise MiIcpError("hipcc not found; cannot build libmi_icp.so")
    os.makedirs(obj_dir, exist_ok=True)
    import fcntl
    import hashlib
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not variant and not needs_build():
                return LIB_PATH          # (another process built it while this one waited)
            return _build_locked(hipcc, lib_path, obj_dir, force, verbose, tuple(defines), hashlib)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(hipcc, lib_path, obj_dir, force, verbose, defines, hashlib):
    headers = [s for s in _sources() if s.endswith(".h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    flags_hash = hashlib.sha1(" ".join(HIPCC_FLAGS + list(defines)).encode()).hexdigest()
    stamp = os.path.join(obj_dir, "flags.stamp")
    if not (os.path.exists(stamp) and open(stamp).read().strip() == flags_hash):
        force = True                     # objects of another flag set: none may be reused

    def compile_unit(u):
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(obj_dir, u + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(newest_header, os.path.getmtime(src)):
            return obj
        tmp = "%s.tmp.%d" % (obj, os.getpid())
        cmd = [hipcc] + HIPCC_FLAGS + list(defines) + ["-c", src, "-o", tmp]
        if verbose:
            print(" ".join(cmd[:-1] + [obj]), flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, obj)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(UNITS)) as pool:
        objs = list(pool.map(compile_unit, UNITS))
    with open(stamp + ".tmp", "w") as f:
        f.write(flags_hash + "\n")
    os.replace(stamp + ".tmp", stamp)
    tmp = "%s.tmp.%d" % (lib_path, os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, lib_path)
    return lib_path


ITERATION_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_float, C.c_float)   # mi_icp_iteration_fn


class Params(C.Structure):
    _fields_ = [("relative_fitness", C.c_float), ("relative_rmse", C.c_float),
                ("max_iteration", C.c_int32), ("det_thresh", C.c_float)]


class Result(C.Structure):
    _fields_ = [("transformation", C.c_float * 16), ("fitness", C.c_float),
                ("inlier_rmse", C.c_float), ("n_correspondences", C.c_int64),
                ("iterations", C.c_int32), ("nn_passes", C.c_int32)]


class OdometryOption(C.Structure):
    _fields_ = [("num_levels", C.c_int32), ("iterations", C.c_int32 * 8), ("max_depth_diff", C.c_float),
                ("min_depth", C.c_float), ("max_depth", C.c_float), ("nu", C.c_float), ("sigma2_init", C.c_float),
                ("inv_sigma_mat_diag", C.c_float * 6)]


# name -> (restype, argtypes); must list every MI_ICP_API symbol of include/mi_icp.h
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SIGNATURES = {
    "mi_icp_create": (_I, [_I, C.POINTER(_P)]),
    "mi_icp_destroy": (None, [_P]),
    "mi_icp_last_error": (C.c_char_p, [_P]),
    "mi_icp_version": (C.c_char_p, []),
    "mi_icp_set_stream": (_I, [_P, _P]),
    "mi_icp_synchronize": (_I, [_P]),
    "mi_icp_set_target": (_I, [_P, _P, _P, _P, _L, _I]),
    "mi_icp_set_source": (_I, [_P, _P, _P, _P, _L, _I]),
    "mi_icp_search_radius_1nn": (_I, [_P, _P, _F, _P, _P, _I, _P]),
    "mi_icp_get_correspondences": (_I, [_P, _P, _L, C.POINTER(_L), _I]),
    "mi_icp_set_correspondences": (_I, [_P, _P, _L, _I]),
    "mi_icp_compute_system": (_I, [_P, _I, _P, _P]),
    "mi_icp_compute_transformation": (_I, [_P, _I, _P, _F, _P]),
    "mi_icp_compute_rmse": (_I, [_P, _I, _P, C.POINTER(_F)]),
    "mi_icp_solve_system": (_I, [_P, _F, _P]),
    "mi_icp_kabsch_from_sums": (_I, [_P, _L, _P]),
    "mi_icp_vector6_to_matrix4": (None, [_P, _P]),
    "mi_icp_lzf_decompress": (_L, [_P, _L, _P, _L]),
    "mi_icp_lzf_compress": (_L, [_P, _L, _P, _L]),
    "mi_icp_evaluate_registration": (_I, [_P, _F, _P, C.POINTER(Result)]),
    "mi_icp_registration_icp": (_I, [_P, _I, _F, _P, C.POINTER(Params), C.POINTER(Result)]),
    "mi_icp_icp_begin": (_I, [_P, _I, _F, _P, _F, C.POINTER(Result)]),
    "mi_icp_icp_iterate": (_I, [_P, _I, C.POINTER(Result)]),
    "mi_icp_transform": (_I, [_P, _P, _P, _P, _P, _L, _I]),
    "mi_icp_compute_bounds": (_I, [_P, _P, _L, _I, _P, _P, _P]),
    "mi_icp_affine": (_I, [_P, _P, _F, _I, _P, _P, _P, _P, _P, _L, _I]),
    "mi_icp_voxel_downsample": (_I, [_P, _P, _P, _P, _L, _F, _P, _P, _P, C.POINTER(_L), _I]),
    "mi_icp_create_from_depth": (_I, [_P, _P, _I, _P, _I, _I, _I, _P, _P, _F, _F, _F, _I, _I, _I, _I,
                                      _P, _P, _P, C.POINTER(_L), _I]),
    "mi_icp_compute_rgbd_odometry": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P, C.POINTER(_I), _P, _P, _I]),
    "mi_icp_compute_weighted_rgbd_odometry": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, C.POINTER(_I), _P, _P, _P, _I]),
    "mi_icp_covariances_from_normals": (_I, [_P, _P, _L, _F, _P, _I]),
    "mi_icp_estimate_normals_knn": (_I, [_P, _P, _L, _I, _P, _I]),
    "mi_icp_estimate_normals_radius": (_I, [_P, _P, _L, _F, _I, _P, _I]),
    "mi_icp_search_knn": (_I, [_P, _P, _L, _I, _F, _P, _P, C.POINTER(_L), _I]),
    "mi_icp_set_target_colors": (_I, [_P, _P, _I]),
    "mi_icp_set_source_colors": (_I, [_P, _P, _I]),
    "mi_icp_set_lambda_geometric": (_I, [_P, _F]),
    "mi_icp_compute_color_gradients": (_I, [_P, _F, _I, _P, _I]),
    "mi_icp_registration_colored_icp": (_I, [_P, _F, _P, C.POINTER(Params), _F, C.POINTER(Result)]),
    "mi_icp_comm_unique_id": (_I, [_P]),
    "mi_icp_comm_init": (_I, [_P, _P, _I, _I]),
    "mi_icp_comm_init_local": (_I, [_P, C.c_char_p, _I, _I]),
    "mi_icp_comm_kind": (_I, [_P]),
    "mi_icp_comm_autotune": (_I, [_P, _I, _P, _P]),
    "mi_icp_comm_destroy": (_I, [_P]),
    "mi_icp_set_global_source_count": (_I, [_P, _L]),
    "mi_icp_spatial_order": (_I, [_P, _P, _L, _P, _I]),
    "mi_icp_set_iteration_callback": (_I, [_P, _P, _P]),
    "mi_icp_set_profiling": (_I, [_P, _I]),
    "mi_icp_get_profile": (_I, [_P, _P]),
    # include/mi_icp_debug.h (test-only)
    "mi_icp_debug_sort_pairs": (_I, [_P, _P, _P, _L, _I]),
    "mi_icp_debug_exclusive_scan": (_I, [_P, _P, _P, _L, _P]),
    "mi_icp_debug_morton_order": (_I, [_P, _P, _L, _P]),
    "mi_icp_debug_nn_stats": (_I, [_P, _P, _F, _I, _P]),
    "mi_icp_debug_nn_stats8": (_I, [_P, _P, _F, _I, _P]),
    "mi_icp_debug_get_tree": (_I, [_P, C.POINTER(_L), _P, _P]),
    "mi_icp_debug_drop_seeds": (_I, [_P]),
    "mi_icp_debug_last_search_kind": (_I, [_P]),
    "mi_icp_debug_last_voxel_path": (_I, [_P]),
    "mi_icp_debug_occupancy": (_I, [_I]),
    "mi_icp_debug_loop_counters": (_I, [_P, _P]),
    "mi_icp_debug_locate": (_I, [_P, _P, _P]),
    "mi_icp_debug_set_step_stamps": (_I, [_P, _I]),
    "mi_icp_debug_get_step_stamps": (_I, [_P, _P, C.POINTER(C.c_double)]),
    "mi_icp_debug_solve_both": (_I, [_I, _P, _I, C.c_float, _P, _P, _P, _P]),
    "mi_icp_debug_get_leaf_regions": (_I, [_P, _P]),
    "mi_icp_debug_get_leaf_halos": (_I, [_P, _P]),
    "mi_icp_debug_eigen3": (_I, [_I, _P, _L, _P, _P, _P]),
}

_lib = None


def load():
    """Load libmi_icp.so and bind every entry point; raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    override = os.environ.get("MI_ICP_LIB_PATH")   # A/B runs of two builds on one box (scripts/gpu_ab_libs.sh)
    path = override or LIB_PATH
    if not os.path.exists(path):
        raise MiIcpError(
            "libmi_icp.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or cupoch_amd._lib.build(); there is no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError as e:  # e.g. libamdhip64 missing
        raise MiIcpError("cannot load %s: %s" % (path, e))
    for name, (res, args) in SIGNATURES.items():
        if override and name.startswith("mi_icp_debug_") and not hasattr(lib, name):
            continue     # (an older build under comparison may lack a test-only entry point)
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
