import ctypes as C, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cupoch_amd.engine import Engine
n = 1_000_000
rng = np.random.default_rng(1)
pts = rng.random((n, 3), dtype=np.float32)
s = n ** (-1 / 3)
e = Engine(0)
e.set_target(torch.from_numpy(pts).cuda())
info = (C.c_int64 * 5)()
e._chk(e._L.mi_icp_debug_get_tree(e._ctx, info, None, None))
nleaf = int(info[1])
halos = np.empty((nleaf, 29, 32), np.float32)
e._chk(e._L.mi_icp_debug_get_leaf_halos(e._ctx, halos.ctypes.data_as(C.c_void_p)))
reg = np.empty((nleaf, 8), np.float32)
e._chk(e._L.mi_icp_debug_get_leaf_regions(e._ctx, reg.ctypes.data_as(C.c_void_p)))
ok = reg[:, 7] > 0
print("leaves", nleaf, "with halo", ok.sum())
h = halos[ok]
ext = h[:, :18, 31].copy().view(np.int32)
r1 = h[:, :18, 7]
r2 = np.take_along_axis(h[:, :, 7], np.where(ext >= 0, ext, 0), 1)
r = np.where(ext >= 0, r2, r1) / s
print("near reaches/s", np.quantile(h[:, 26:29, 7] / s, [0.1, 0.5, 0.9], axis=0).round(2).tolist())
print("extension lines per leaf", np.bincount((ext >= 0).sum(1), minlength=9))
print("face reach/s quantiles per line", np.quantile(r[:, :6], [0.01, 0.1, 0.5, 0.9, 0.99]))
print("edge reach/s quantiles per line", np.quantile(r[:, 6:], [0.01, 0.1, 0.5, 0.9, 0.99]))
print("leaf min reach/s", np.quantile(r.min(1), [0.001, 0.01, 0.1, 0.5, 0.9]))
ext = (reg[ok, 4:7] - reg[ok, 0:3]) / s
fin = np.isfinite(ext).all(1)
print("region extent/s", np.quantile(ext[fin], [0.1, 0.5, 0.9]), "delta0/s", np.quantile(reg[ok, 7] / s, [0.1, 0.5, 0.9]))
used = (halos[ok][:, :18, 24:31].copy().view(np.int32) >= 0).sum(2)
print("points per face line", np.bincount(used[:, :6].ravel(), minlength=8), "edge", np.bincount(used[:, 6:].ravel(), minlength=8))
