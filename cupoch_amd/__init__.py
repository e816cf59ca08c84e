"""cupoch_amd -- the MI355X-native ICP registration path of cupoch.

Drop-in for the hot path only (registration_icp and friends over
geometry.PointCloud); see DESIGN.md for scope and INTEGRATION.md for how it
binds behind cupoch's C++ / Python surface.  Compute lives in
cupoch_amd/lib/libmi_icp.so (HIP, gfx950); importing this package does not need
a GPU, calling into it does."""
from . import _lib, utility                                    # noqa: F401
from ._lib import MiIcpError, build                           # noqa: F401
from .utility import initialize_allocator                     # noqa: F401

__all__ = ["geometry", "registration", "utility", "io", "engine", "distributed", "camera", "kinfu", "odometry", "build",
           "initialize_allocator", "MiIcpError"]


def __getattr__(name):
    if name in ("geometry", "registration", "io", "engine", "distributed", "camera", "kinfu", "odometry"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
