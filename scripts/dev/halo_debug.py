import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from cupoch_amd.engine import Engine
from test_gpu_tree_invariants import get_tree, halo_reaches
rng = np.random.default_rng(3)
clustered = np.concatenate([c + rng.normal(0, 0.003, (s, 3)).astype(np.float32)
                            for c, s in zip(rng.random((40, 3)).astype(np.float32), rng.integers(5, 3000, 40))])
planar = rng.random((30000, 3), dtype=np.float32); planar[:, 2] = 0.25
grid = np.round(rng.random((40000, 3)) * 32).astype(np.float32) / 32
e = Engine(0)
for name, pts in (("clustered", clustered), ("planar", planar), ("grid", grid)):
    e.set_target(pts)
    nts, nleaf, leaf_first, rec, lines, nt = get_tree(e)
    xyz = np.stack([lines[:, 0:8], lines[:, 8:16], lines[:, 16:24]], -1).reshape(-1, 3)
    orig = lines[:, 24:32].copy().view(np.int32).reshape(-1)
    finite = np.isfinite(xyz).all(1) & (orig >= 0)
    halos = np.empty((nleaf, 8, 32), np.float32)
    e._chk(e._L.mi_icp_debug_get_leaf_halos(e._ctx, halos.ctypes.data_as(C.c_void_p)))
    reg = np.empty((nleaf, 8), np.float32)
    e._chk(e._L.mi_icp_debug_get_leaf_regions(e._ctx, reg.ctypes.data_as(C.c_void_p)))
    lo, hi = reg[:, 0:3], reg[:, 4:7]
    wa, wb = reg[:, 3].copy().view(np.uint32), reg[:, 7].copy().view(np.uint32)
    have = np.flatnonzero(wb != 0)
    leaf_of = np.arange(len(xyz)) // 8
    p64 = xyz.astype(np.float64)
    bad = 0
    for L in have:
        foreign = finite & (leaf_of != L)
        up = p64 - hi[L].astype(np.float64); dn = lo[L].astype(np.float64) - p64
        with np.errstate(invalid="ignore"):
            dist = np.maximum(np.maximum(up, dn).max(1), 0.0)
        reaches = halo_reaches(wa[L], wb[L])
        slots = halos[L, :, 24:32].copy().view(np.int32)
        seen = set()
        for k in range(8):
            seen |= set(slots[k][slots[k] >= 0].tolist())
            must = np.flatnonzero(foreign & (dist < float(reaches[k])))
            miss = set(must.tolist()) - seen
            if miss:
                bad += 1
                if bad <= 3:
                    bound = np.array([wb[L] & 0xffff0000], np.uint32).view(np.float32)[0]
                    n_in = int((foreign & (dist < bound)).sum())
                    print(name, "leaf", L, "ring", k, "reach", reaches[k], "bound16", bound, "foreign within bound", n_in, "missing", len(miss),
                          "missing dists", np.sort(dist[list(miss)])[:4], "ring dists", np.sort(dist[slots[k][slots[k] >= 0]]),
                          "region", lo[L], hi[L], "missing leaves", sorted(set((np.array(list(miss)) // 8).tolist()))[:8])
                break
    print(name, "leaves with halo", len(have), "bad", bad)
