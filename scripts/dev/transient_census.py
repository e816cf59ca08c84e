"""Census of the seeded search at several points of the bench's transient (10M noisy points, started 1.5 spacings off):
records / leaf batches / halo lines per packet, packets that walk, lanes unfinished when the walk starts."""
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
os.environ.setdefault("MI_ICP_CENSUS_WHY", "1")
for k in (1, 2, 4, 8, 12, 16, 24):
    eng.set_target(d_tgt, d_nrm)
    eng.set_source(d_noisy)
    eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, init, -1.0)
    res = eng.icp_iterate(k)
    T = np.ascontiguousarray(np.array(res.transformation, np.float32).reshape(4, 4))   # column-major as the ABI takes it
    out = (C.c_uint64 * 8)()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng._chk(eng._L.mi_icp_debug_nn_stats8(eng._ctx, T.ctypes.data_as(C.c_void_p), float(max_dist), 1, out))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    o = list(out)
    p = max(o[2], 1)
    print("after %2d iterations (rmse %.3f spacings): records/packet %.2f, leaf batches/packet %.2f, halo lines/packet %.1f, packets with a halo phase %.1f %%, "
          "packets that walk %.1f %%, lanes unfinished at the walk %.1f/packet, slowest packet %d steps; census search %.2f ms"
          % (k, res.inlier_rmse / s, o[0] / p, o[1] / p, o[4] / p, 100.0 * o[5] / p, 100.0 * o[6] / p, o[7] / p, o[3], dt * 1e3), flush=True)
