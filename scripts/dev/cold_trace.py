"""The bench's COLD call on its clean clouds (set_target + set_source + 30 iterations, inputs resident): the whole call,
median of 5, and the first searches one by one (HIP events)."""
import os, sys, time, numpy as np, torch
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
mode = "re-location with halos only"
tc, tl = [], []
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.set_target(d_tgt, d_nrm)
    eng.set_source(d_src)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    r = eng.registration_icp(_lib.EST_POINT_TO_PLANE, max_dist, None, 0.0, 0.0, 30, -1.0)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    tc.append((t2 - t0) * 1e3); tl.append((t2 - t1) * 1e3)
T = np.array(r.transformation, np.float32).reshape(4, 4).T
print("%s, n = %d: cold call %.3f ms (median of the last 5: %s), its loop %.3f ms; fitness %.4f rmse %.3g spacings, |T - T_gt|_F %.3g"
      % (mode, n, float(np.median(tc[1:])), " ".join("%.2f" % x for x in tc[1:]), float(np.median(tl[1:])), r.fitness,
         r.inlier_rmse / s, float(np.linalg.norm(T - T_gt))), flush=True)
for rep in range(2):
    eng.set_target(d_tgt, d_nrm)
    eng.set_source(d_src)
    eng.set_profiling(True)
    res = eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, None, -1.0)
    p = eng.get_profile()
    line = ["   stepped, repetition %d: first search %.3f ms (kind %d);" % (rep, p["nn_ms"], eng.last_search_kind())]
    for k in range(6):
        p0 = eng.get_profile()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = eng.icp_iterate(1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        p1 = eng.get_profile()
        line.append("%d: nn %.3f wall %.3f rmse %.2g |" % (k + 2, p1["nn_ms"] - p0["nn_ms"], dt, res.inlier_rmse / s))
    eng.set_profiling(False)
    print(" ".join(line), flush=True)
