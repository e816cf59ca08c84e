"""The bench's transient after k iterations (halos built): the census and the time of the seeded search from the previous
matches against the same search from the leaves the queries FALL INTO (mi_icp_debug_locate) -- is a query's own leaf
the better seed while every match is still wrong?"""
import ctypes as C, os, sys, numpy as np, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd import _lib
from cupoch_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
src, tgt, nrm, T_gt, max_dist = synth(n)
s = n ** (-1.0 / 3.0)
eng = Engine(0)
d_tgt, d_nrm = torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda()
init = np.eye(4, dtype=np.float32)
init[:3, 3] = (1.5 * s / np.sqrt(3.0)) * np.array([1.0, -1.0, 1.0], np.float32)
ang = 0.5 * s
init[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]], np.float32)
rng = np.random.default_rng(6)
noisy_all = (src + rng.normal(0.0, 0.15 * s, src.shape)).astype(np.float32)
d_noisy = torch.from_numpy(np.ascontiguousarray(noisy_all)).cuda()


def census(tag, T):
    out = (C.c_uint64 * 8)()
    eng._chk(eng._L.mi_icp_debug_nn_stats8(eng._ctx, T.ctypes.data_as(C.c_void_p), float(max_dist), 1, out))
    o = list(out); p = max(o[2], 1)
    print("   %s: records/packet %.2f, leaf batches/packet %.2f, halo lines/packet %.1f, packets that walk %.1f %%, lanes unfinished at the walk %.2f/packet"
          % (tag, o[0] / p, o[1] / p, o[4] / p, 100.0 * o[6] / p, o[7] / p), flush=True)


def timed(T):   # one plain seeded search under T (HIP events)
    eng.set_profiling(True)
    p0 = eng.get_profile()
    eng.search_radius_1nn(max_dist, T.T.copy(), want_d2=False)
    p1 = eng.get_profile()
    eng.set_profiling(False)
    return p1["nn_ms"] - p0["nn_ms"]


got = np.empty(n, np.int32)
for k in (2, 6, 12, 16):
    eng.set_target(d_tgt, d_nrm)
    eng.set_source(d_noisy)
    eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, init, -1.0)
    res = eng.icp_iterate(k)
    T = np.ascontiguousarray(np.array(res.transformation, np.float32).reshape(4, 4))   # column-major as the ABI takes it
    print("after %2d iterations (rmse %.3f spacings):" % (k, res.inlier_rmse / s), flush=True)
    census("seeds = the previous matches", T)
    t_stale = timed(T)          # (leaves its matches as seeds: the same state as before)
    t_stale = timed(T)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng._chk(eng._L.mi_icp_debug_locate(eng._ctx, T.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p)))
    census("seeds = the located leaves  ", T)
    eng._chk(eng._L.mi_icp_debug_locate(eng._ctx, T.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p)))
    t_loc = timed(T)
    print("   search from the previous matches %.3f ms, from the located leaves %.3f ms (+ ~0.12 for the descent)" % (t_stale, t_loc), flush=True)
