"""The reference's pybind11 module (`cupoch_pybind`: utility / geometry / registration) built on this
repository's C++ surface -- cupoch_amd/cpp/src/pybind_module.cpp, compiled by `make -C cupoch_amd/cpp`
(`__graft_entry__.build()`).  Usage, as with the reference's package:

    from cupoch_amd import pybind as cph
    pcd = cph.geometry.PointCloud(); pcd.points = cph.utility.Vector3fVector(xyz)
    res = cph.registration.registration_icp(src, tgt, 0.02, np.eye(4, dtype=np.float32),
                                            cph.registration.TransformationEstimationPointToPlane())

The ctypes mirror (cupoch_amd.registration / .geometry) offers the same names without a compiled
module and is what most parity tests drive; this one shows the binding a cupoch maintainer keeps."""
import importlib
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib")


def _load():
    if _LIB not in sys.path:
        sys.path.insert(0, _LIB)
    try:
        return importlib.import_module("cupoch_pybind")
    except ImportError:
        from . import _lib
        _lib.build()                                                 # libmi_icp.so first
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "cpp")])
        importlib.invalidate_caches()
        return importlib.import_module("cupoch_pybind")


_m = _load()
utility, geometry, registration = _m.utility, _m.geometry, _m.registration
initialize_allocator = _m.initialize_allocator
