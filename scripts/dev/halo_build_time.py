"""Kernel time of the halo build on a 10M-point target (hip events around ensure via debug call)."""
import ctypes as C, numpy as np, torch, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cupoch_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts = torch.rand((n, 3), device="cuda")
e = Engine(0)
for it in range(3):
    e.set_target(pts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    src = torch.rand((1000, 3), device="cuda")
    e.set_source(src)
    T = np.eye(4, dtype=np.float32)
    out = (C.c_uint64 * 8)()
    # a seeded pass builds the halos first (ensure_links) -- time = build + a tiny search
    e._chk(e._L.mi_icp_debug_nn_stats8(e._ctx, T.ctypes.data_as(C.c_void_p), 0.01, 0, out))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    e._chk(e._L.mi_icp_debug_nn_stats8(e._ctx, T.ctypes.data_as(C.c_void_p), 0.01, 1, out))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("unseeded small pass %.3f ms, halo build + seeded small pass %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
