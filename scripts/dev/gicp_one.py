"""config 5 (GICP 5M-vs-5M) alone: iteration time and the final transform's error."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd import _lib
from cupoch_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
src, tgt, nrm, T_gt, max_dist = synth(n)
eng = Engine(0)
d_tgt, d_nrm, d_src = torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda(), torch.from_numpy(src).cuda()
src_nrm = (nrm.astype(np.float64) @ np.linalg.inv(T_gt.astype(np.float64))[:3, :3].T).astype(np.float32)
tcov = eng.covariances_from_normals(d_nrm, 1e-3)
scov = eng.covariances_from_normals(torch.from_numpy(src_nrm).cuda(), 1e-3)
eng.set_target(d_tgt, d_nrm, tcov)
eng.set_source(d_src, None, scov)
eng.set_profiling(True)
eng.icp_begin(_lib.EST_GENERALIZED, max_dist, None, -1.0)
eng.icp_iterate(5)
p0 = eng.get_profile(); torch.cuda.synchronize(); t0 = time.perf_counter()
res = eng.icp_iterate(30)
torch.cuda.synchronize(); dt = time.perf_counter() - t0; p1 = eng.get_profile()
T = np.array(res.transformation, np.float32).reshape(4, 4).T
print("gicp %d: %.4f ms/iter  nn %.4f  reduce %.4f  fitness %.4f rmse %.3e T_err %.3e" % (n, dt / 30 * 1e3, (p1["nn_ms"] - p0["nn_ms"]) / 30, (p1["reduce_ms"] - p0["reduce_ms"]) / 30, res.fitness, res.inlier_rmse, np.linalg.norm(T - T_gt)))
