"""The second search of the bench's cold call on its clean clouds, without halos (run with MI_ICP_NO_LINKS=1): the census
of the seeded search from the previous matches against the same search from the leaves the moved queries fall into
(mi_icp_debug_locate), and how many of those leaves hold the query's true partner."""
import ctypes as C, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd import _lib
from cupoch_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
src, tgt, nrm, T_gt, max_dist = synth(n)
s = n ** (-1.0 / 3.0)
eng = Engine(0)
d_tgt, d_nrm, d_src = torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda(), torch.from_numpy(src).cuda()
eng.set_target(d_tgt, d_nrm)
eng.set_source(d_src)
eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, None, -1.0)
res = eng.icp_iterate(1)
T = np.ascontiguousarray(np.array(res.transformation, np.float32).reshape(4, 4))   # after the first step; column-major as the ABI takes it
# the same state again: first pass under the identity (its matches are the seeds), source re-sorted by match
eng.set_source(d_src)
eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, None, -1.0)


def census(tag):
    out = (C.c_uint64 * 8)()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng._chk(eng._L.mi_icp_debug_nn_stats8(eng._ctx, T.ctypes.data_as(C.c_void_p), float(max_dist), 1, out))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    o = list(out); p = max(o[2], 1)
    print("%s: records/packet %.2f, leaf batches/packet %.2f, halo lines/packet %.1f, packets that walk %.1f %%, lanes unfinished "
          "at the walk %.2f/packet, slowest packet %d steps; census search %.2f ms"
          % (tag, o[0] / p, o[1] / p, o[4] / p, 100.0 * o[6] / p, o[7] / p, o[3], dt * 1e3), flush=True)


census("seeds = the first pass's matches (after the first step, rmse %.2g spacings)" % (res.inlier_rmse / s))
# where every target point lives
info = (C.c_int64 * 5)()
eng._chk(eng._L.mi_icp_debug_get_tree(eng._ctx, info, None, None))
nleaf = int(info[1])
lines = np.empty((nleaf, 32), np.float32)
eng._chk(eng._L.mi_icp_debug_get_tree(eng._ctx, info, None, lines.ctypes.data_as(C.c_void_p)))
orig = lines[:, 24:32].copy().view(np.int32).reshape(-1)
real = orig >= 0
lives = np.empty(n, np.int64)
lives[orig[real]] = np.nonzero(real)[0] >> 3
# source point i is target point perm[i] moved: bench.synth permutes with seed 44
perm = np.random.Generator(np.random.PCG64(44)).permutation(n)
got = np.empty(n, np.int32)
eng._chk(eng._L.mi_icp_debug_locate(eng._ctx, T.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p)))
same = got == lives[perm]
print("located leaves that hold the query's partner: %.3f %% (of %d)" % (100.0 * same.mean(), n), flush=True)
census("seeds = the located leaves")
idx, d2, _ = eng.search_radius_1nn(max_dist, T.T.copy())
print("matches equal the partners: %.4f %%" % (100.0 * (idx == perm).mean()))
