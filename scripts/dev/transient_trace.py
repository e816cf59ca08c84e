"""The bench's transient, iteration by iteration: search time (HIP events) and how many re-locations have run -- one line per
iteration.  MI_ICP_NO_LOCATE_PLANES=1 gives the former form (stale seeds, greedy first descent) for comparison."""
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
d_tgt, d_nrm = torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda()
init = np.eye(4, dtype=np.float32)
init[:3, 3] = (1.5 * s / np.sqrt(3.0)) * np.array([1.0, -1.0, 1.0], np.float32)
ang = 0.5 * s
init[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]], np.float32)
rng = np.random.default_rng(6)
noisy_all = (src + rng.normal(0.0, 0.15 * s, src.shape)).astype(np.float32)
d_noisy = torch.from_numpy(np.ascontiguousarray(noisy_all)).cuda()
mode = "former form (MI_ICP_NO_LOCATE_PLANES)" if os.environ.get("MI_ICP_NO_LOCATE_PLANES") else "locate by planes + re-location"
for rep in range(2):      # (the first repetition builds the halos inside the loop; the second -- a context that has asked before -- ahead of it)
    eng.set_target(d_tgt, d_nrm)
    eng.set_source(d_noisy)
    eng.set_profiling(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, init, -1.0)
    torch.cuda.synchronize(); tb = (time.perf_counter() - t0) * 1e3
    p = eng.get_profile()
    line = ["%s, repetition %d: begin %.2f ms (first search %.3f ms, kind %d);" % (mode, rep, tb, p["nn_ms"], eng.last_search_kind())]
    total = tb
    for k in range(30):
        p0 = eng.get_profile()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = eng.icp_iterate(1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        total += dt
        p1 = eng.get_profile()
        line.append("%d: nn %.3f wall %.3f rmse %.3f |" % (k + 1, p1["nn_ms"] - p0["nn_ms"], dt, res.inlier_rmse / s))
    eng.set_profiling(False)
    print(" ".join(line))
    print("   sum of the stepped calls %.2f ms" % total, flush=True)
